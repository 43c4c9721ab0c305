"""Stand-alone per-kernel efficiency at a given batch size (serial=True: one stream; profiling mode 2)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from deepsignal_amd import synth, weights as W
from deepsignal_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
e = Engine(max_batch=B, slots=1, serial=True); e.load_weights(W.random_weights(seed=1))
f = synth.synthetic_features(B, seed=2)
args = [f[k] for k in ("kmer", "means", "stds", "sanums", "signals")]
e.run(*args)
e.set_profiling(2); e.reset_stage_times()
for _ in range(3): e.run(*args)
e.sync()
tot = 0
for k in e.kernel_stats():
    if k["launches"]:
        us = 1e3 * k["total_ms"] / 3; tot += us
        print("%-30s %3d launches %9.1f us  %6.1f TFLOP/s" % (k["name"], k["launches"] // 3, us, k["flops"] / (k["total_ms"] * 1e-3) / 1e12 if k["flops"] else 0))
print("batch %d: sum %.1f us -> %.0f sites/s serial" % (B, tot, B / tot * 1e6))
