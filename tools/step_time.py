"""Whole-step throughput of one engine configuration (resident inputs, pipelined slots), median of several windows.
usage: python tools/step_time.py <precision> <batch> [steps=40] [windows=5] [key=value engine options]"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from deepsignal_amd import synth, weights as W
from deepsignal_amd.engine import Engine
prec, B = sys.argv[1], int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
wins = int(sys.argv[4]) if len(sys.argv) > 4 else 5
opts = {}
for kv in sys.argv[5:]:
    k, v = kv.split("=")
    opts[k] = {"true": True, "false": False}.get(v.lower(), int(v) if v.lstrip("-").isdigit() else v)
dev = torch.device("cuda", 0)
f = synth.synthetic_features(B, seed=1)
keys = ("kmer", "means", "stds", "sanums", "signals")
d = {k: torch.from_numpy(f[k]).to(dev) for k in keys}
e = Engine(device=0, max_batch=B, precision=prec, **opts)
e.load_weights(W.random_weights(seed=W.WEIGHT_SEED))
act = torch.zeros((B, 2), dtype=torch.float32, device=dev); pred = torch.zeros((B,), dtype=torch.int32, device=dev)
step = lambda: e.run_device(B, *(d[k].data_ptr() for k in keys), act.data_ptr(), pred.data_ptr())
for _ in range(10): step()
e.sync()
rates = []
for w in range(wins):
    t0 = time.perf_counter()
    for _ in range(steps): step()
    e.sync()
    rates.append(steps * B / (time.perf_counter() - t0))
rates.sort()
print(json.dumps({"precision": prec, "batch": B, "options": opts, "sites_per_s_median": round(rates[len(rates) // 2], 1),
                  "min": round(rates[0], 1), "max": round(rates[-1], 1), "ms_per_step": round(1e3 * B / rates[len(rates) // 2], 4)}))
e.close()
