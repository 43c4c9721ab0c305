"""Soak test: minutes of randomly sized forwards through all three boundaries (blocking, submit/wait, device pointers),
every result compared bit for bit with a reference pass over the same sites. Exercises the slot rotation, the bounded
plan cache (hundreds of distinct sizes), graph capture for recurring sizes and the ragged-tail kernels.

usage: python tools/soak.py [seconds] [precision] [max_batch] [lstm_tiling] [threestep]
(with an lstm_tiling override the reference pass comes from a default engine: every tiling must give its bits)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from deepsignal_amd import synth, weights as W
from deepsignal_amd.engine import Engine

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 512          # max_batch of the engine under test
POOL = max(8192, 4 * B)
keys = ("kmer", "means", "stds", "sanums", "signals")
feats = synth.synthetic_features(POOL, seed=77)
tiling = sys.argv[4] if len(sys.argv) > 4 else "auto"
extra = {"fold_fc": False} if len(sys.argv) > 5 and sys.argv[5] == "threestep" else {}      # the reference graph's three-step joint model
w = W.random_weights(seed=5, lstm_bias_std=0.1)
if tiling != "auto":
    ref_eng = Engine(max_batch=B, precision=prec, **extra)
    ref_eng.load_weights(w)
    ref_act, ref_pred = ref_eng.run(*(feats[k] for k in keys))
    ref_eng.close()
eng = Engine(max_batch=B, precision=prec, lstm_tiling=tiling, **extra)
eng.load_weights(w)
if tiling == "auto":
    ref_act, ref_pred = eng.run(*(feats[k] for k in keys))
dev = torch.device("cuda", 0)
d = {k: torch.from_numpy(feats[k]).to(dev) for k in keys}
rng = np.random.default_rng(1)
t0 = time.time(); it = 0; sites = 0; bad = 0
pending = []
while time.time() - t0 < secs:
    mode = it % 3
    if mode == 0:                                   # blocking host call, possibly several passes
        n = int(rng.integers(1, 3 * B)); s = int(rng.integers(0, POOL - n))
        a, p = eng.run(*(feats[k][s:s + n] for k in keys))
        bad += int(not (np.array_equal(a, ref_act[s:s + n]) and np.array_equal(p, ref_pred[s:s + n])))
        sites += n
    elif mode == 1:                                 # asynchronous host boundary, `slots` deep
        for _ in range(eng.slots + 3):
            n = int(rng.integers(1, B + 1)); s = int(rng.integers(0, POOL - n))
            if len(pending) == eng.slots:
                t, s0, n0 = pending.pop(0)
                a, p = eng.wait(t)
                bad += int(not (np.array_equal(a, ref_act[s0:s0 + n0]) and np.array_equal(p, ref_pred[s0:s0 + n0])))
            pending.append((eng.submit(*(feats[k][s:s + n] for k in keys)), s, n)); sites += n
        while pending:
            t, s0, n0 = pending.pop(0)
            a, p = eng.wait(t)
            bad += int(not (np.array_equal(a, ref_act[s0:s0 + n0]) and np.array_equal(p, ref_pred[s0:s0 + n0])))
    else:                                           # device pointers, pipelined over the slots
        outs = []
        for _ in range(12):
            n = int(rng.choice([B, B, B, int(rng.integers(1, B + 1))])); s = int(rng.integers(0, POOL - n))
            oa = torch.empty((n, 2), dtype=torch.float32, device=dev); op = torch.empty((n,), dtype=torch.int32, device=dev)
            eng.run_device(n, *(d[k][s:s + n].data_ptr() for k in keys), oa.data_ptr(), op.data_ptr())
            outs.append((oa, op, s, n)); sites += n
        eng.sync()
        for oa, op, s, n in outs:
            bad += int(not (np.array_equal(oa.cpu().numpy(), ref_act[s:s + n]) and np.array_equal(op.cpu().numpy(), ref_pred[s:s + n])))
    it += 1
eng.close()
print("soak %s (max_batch %d, lstm_tiling %s): %.0f s, %d rounds, %d sites, %d mismatching calls" % (prec, B, tiling, time.time() - t0, it, sites, bad))
sys.exit(1 if bad else 0)
