import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from deepsignal_amd import synth, weights as W
from deepsignal_amd.engine import Engine
w = W.random_weights(seed=5, lstm_bias_std=0.1)
keys = ("kmer", "means", "stds", "sanums", "signals")
N = 16384
f = synth.synthetic_features(N, seed=9)
for prec in ("fp32", "bf16_all", "bf16"):
    ref = Engine(max_batch=512, precision=prec); ref.load_weights(w)
    ra, rp = ref.run(*(f[k] for k in keys)); ref.close()
    for B in (8192, 16384):
        e = Engine(max_batch=B, precision=prec); e.load_weights(w)
        a, p = e.run(*(f[k] for k in keys))
        a2, p2 = e.run(*(f[k][:B - 37] for k in keys))
        print(prec, B, "equal bits vs batch 512:", bool(np.array_equal(a, ra) and np.array_equal(p, rp)), bool(np.array_equal(a2, ra[:B - 37])), flush=True)
        e.close()
