"""First-contact check of a precision mode against the CPU oracle, tap by tap (debug engine, benign and stress weights), then
stand-alone kernel times next to fp32. usage: python tools/split_check.py [precision=bf16x3] [n=130]"""
import json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from deepsignal_amd import synth, weights as W
from deepsignal_amd.engine import Engine
from oracle import oracle

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 130
KEYS = ("kmer", "means", "stds", "sanums", "signals")
g = np.load("tests/golden/stress_golden.npz")
sets = {"benign": W.random_weights(seed=7, lstm_bias_std=0.1),
        "stress": W.stress_weights(int(g["stress_seed"]), head=g["stress_head"])}
for name, w in sets.items():
    feats = synth.synthetic_features(n, seed=900 + n)
    a32, p32, t32 = oracle.forward(w, feats, "f32", taps=True)
    a64, p64, t64 = oracle.forward(w, feats, "f64", taps=True)
    for p in ("fp32", prec):
        eng = Engine(max_batch=max(160, n), debug=True, precision=p)
        eng.load_weights(w)
        act, pred = eng.run(*(feats[k] for k in KEYS))
        print("== %s / %s: max|d act| vs f64 %.3e (oracle f32: %.3e), vs oracle f32 %.3e, label mismatches %d" % (
            name, p, np.abs(act - a64).max(), np.abs(a32 - a64).max(), np.abs(act - a32).max(), int((pred != p64).sum())))
        for k, ref in t64.items():
            if not (k.startswith("module") or k.startswith("stem") or k in ("signal_feat", "fc1", "logits")):
                continue
            got = eng.intermediate(k, ref.shape)
            sc = max(1.0, float(np.abs(ref).max()))
            print("   %-12s scale %9.3f  hip-f64 %.3e  oracle32-f64 %.3e   (rel %.2e / %.2e)" % (
                k, sc, np.abs(got - ref).max(), np.abs(t32[k] - ref).max(), np.abs(got - ref).max() / sc, np.abs(t32[k] - ref).max() / sc))
        eng.close()
