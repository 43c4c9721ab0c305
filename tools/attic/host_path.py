"""PCIe-inclusive rate of the host-buffer boundary (ds_forward: pageable H2D + forward + D2H, blocking)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from deepsignal_amd import synth, weights as W
from deepsignal_amd.engine import Engine
e = Engine(max_batch=512); e.load_weights(W.random_weights(seed=1))
for n in (512, 4096, 32768):
    f = synth.synthetic_features(n, seed=2)
    args = [f[k] for k in ("kmer", "means", "stds", "sanums", "signals")]
    e.run(*[a[:512] for a in args])
    t0 = time.perf_counter(); reps = max(1, 65536 // n)
    for _ in range(reps): e.run(*args)
    dt = time.perf_counter() - t0
    print("ds_forward host buffers, n=%d per call (looped in 512-site passes): %.0f sites/s" % (n, reps * n / dt))
