"""Eager SERIAL forwards of one engine configuration (every launch on one stream, no graph): the program to run under
rocprofv3 (--pmc passes; --kernel-trace --stats for the stand-alone kernel durations bench.py's roofline quotes).
usage: probe_engine.py <precision> <batch> [threestep] [reps=N]
(threestep: the reference graph's three-step joint model, Engine(fold_fc=False), as bench.py's headline runs it; N forwards, default 2)"""
import os, sys
sys.path.insert(0, os.getcwd())
from deepsignal_amd import synth, weights as W
from deepsignal_amd.engine import Engine
prec, B = sys.argv[1], int(sys.argv[2])
e = Engine(max_batch=B, precision=prec, slots=1, serial=True, fold_fc="threestep" not in sys.argv[3:]); e.load_weights(W.random_weights(seed=1))
e.set_graph(False)
f = synth.synthetic_features(B, seed=2)
args = [f[k] for k in ("kmer", "means", "stds", "sanums", "signals")]
reps = [int(a[5:]) for a in sys.argv[3:] if a.startswith("reps=")]
for _ in range(reps[0] if reps else 2): e.run(*args)
e.close()
