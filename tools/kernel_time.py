"""Stand-alone kernel times of one engine configuration (every launch on one stream, HIP events per run of same-kernel
launches): median and min over several repeats, to compare kernel variants at a +-1 % level.
usage: python tools/kernel_time.py <precision> <batch> [steps=10] [repeats=7] [name filter] [key=value engine options]"""
import json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from deepsignal_amd import synth, weights as W
from deepsignal_amd.engine import Engine

prec, B = sys.argv[1], int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 7
flt = sys.argv[5] if len(sys.argv) > 5 else ""
opts = {}
for kv in sys.argv[6:]:
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
for _ in range(5): step()
e.sync()
e.set_profiling(3)
runs = {}
for r in range(reps):
    e.reset_stage_times()
    for _ in range(steps): step()
    e.sync()
    for k in e.kernel_stats():
        if k["launches"] and flt in k["name"]:
            runs.setdefault(k["name"], []).append(1e3 * k["total_ms"] / steps)
out = {n: {"median_us_per_step": round(float(np.median(v)), 1), "min_us_per_step": round(min(v), 1), "max_us_per_step": round(max(v), 1)}
       for n, v in runs.items()}
print(json.dumps({"precision": prec, "batch": B, "steps": steps, "repeats": reps, "options": opts, "kernels": out}))
e.close()
