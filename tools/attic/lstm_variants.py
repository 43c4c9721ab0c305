"""Stand-alone time of the BiLSTM diagonal launches for every fp32 cell-kernel variant (one stream, profiling mode 1) and
bit-equality of their outputs.  usage: lstm_variants.py [batch]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from deepsignal_amd import synth, weights as W
from deepsignal_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
w = W.random_weights(seed=1)
f = synth.synthetic_features(B, seed=2)
args = [f[k] for k in ("kmer", "means", "stds", "sanums", "signals")]
ref = None
for rep in range(2):
    for var in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("narrow", "wide", "lds1", "lds2")):
        e = Engine(max_batch=B, slots=1, serial=True, lstm_tiling=var); e.load_weights(w)
        act, pred = e.run(*args)
        if ref is None: ref = act
        same = np.array_equal(act, ref)
        e.set_profiling(1); e.reset_stage_times()
        for _ in range(5): e.run(*args)
        e.sync()
        for k in e.kernel_stats():
            if k["launches"] and "lstm" in k["name"]:
                us = 1e3 * k["total_ms"] / k["launches"]
                print("%-6s %-26s %6.2f us/launch  %6.1f TFLOP/s (executed)  bits==narrow: %s" % (var, k["name"], us, k["flops"] / (k["total_ms"] * 1e-3) / 1e12, same))
        e.close()
