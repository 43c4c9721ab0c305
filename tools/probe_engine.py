"""Two eager forwards of one engine configuration (for rocprofv3 --pmc runs).  usage: probe_engine.py <precision> <batch> [threestep]
(threestep: the reference graph's three-step joint model, Engine(fold_fc=False), as bench.py's headline runs it)"""
import os, sys
sys.path.insert(0, os.getcwd())
from deepsignal_amd import synth, weights as W
from deepsignal_amd.engine import Engine
prec, B = sys.argv[1], int(sys.argv[2])
e = Engine(max_batch=B, precision=prec, slots=1, serial=True, fold_fc="threestep" not in sys.argv[3:]); e.load_weights(W.random_weights(seed=1))
e.set_graph(False)
f = synth.synthetic_features(B, seed=2)
args = [f[k] for k in ("kmer", "means", "stds", "sanums", "signals")]
for _ in range(2): e.run(*args)
e.close()
