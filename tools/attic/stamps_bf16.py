import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from deepsignal_amd import synth, weights as W
from deepsignal_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024     # stamp buffer holds 1024 workgroups per module
w = W.random_weights(seed=1)
e = Engine(max_batch=B, precision="bf16_all", slots=1, serial=True, debug_stamps=True); e.load_weights(w)
f = synth.synthetic_features(B, seed=2)
args = [f[k] for k in ("kmer", "means", "stds", "sanums", "signals")]
for _ in range(3): e.run(*args)
names = ["(n)", "prologue(+load)", "P1 mfma", "sync+epi+sync", "P2a", "sync", "P2b", "sync+tail st+sync(+rows out)"]
for m in range(1, 12):
    st = e.intermediate("stamps%d" % m, (16,))
    print("module", m, "wgs", int(st[0]))
    for wv in range(2):
        print("   wave", (0, 7)[wv], " ".join("%s=%.0f" % (names[i], st[wv * 8 + i]) for i in range(1, 8)), "total=%.0f ticks (100 MHz)" % st[wv*8+1:wv*8+8].sum())
