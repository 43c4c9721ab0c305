import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from deepsignal_amd import synth, weights as W
from deepsignal_amd.engine import Engine
"""In-kernel phase stamps of the fused inception kernels. usage: python tools/stamps.py [precision=fp32] [batch=512]"""
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
w = W.random_weights(seed=1)
e = Engine(max_batch=B, serial=True, debug_stamps=True, precision=prec); e.load_weights(w)
f = synth.synthetic_features(B, seed=2)
args = [f[k] for k in ("kmer", "means", "stds", "sanums", "signals")]
for _ in range(3): e.run(*args)
names = ["(n)", "zero+stage0", "sync", "P1epi", "sync", "P2a", "sync", "P2b"]
for m in (2, 5, 10):
    st = e.intermediate("stamps%d" % m, (16,))
    print("module", m, "wgs", int(st[0]))
    for wv in range(2):
        print("   wave", (0, 7)[wv], " ".join("%s=%.0f" % (names[i], st[wv * 8 + i]) for i in range(1, 8)), "total=%.0f cyc(100MHz ticks?)" % st[wv*8+1:wv*8+8].sum())
