"""BASELINE.json configs[2]: 1xMI355X, bf16 conv+FC with fp32 BiLSTM, batch 4096 -- throughput next to the fp32
engine at the same batch, the bf16-vs-fp32 output distance, and per-kernel times (one event pair per run of
same-kernel launches). Prints one JSON object.

usage: python tools/config3.py [batch] [out.json]
"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from deepsignal_amd import synth, spec, weights as W
from deepsignal_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
w = W.random_weights(seed=W.WEIGHT_SEED)
dev = torch.device("cuda", 0)
f = synth.synthetic_features(B, seed=1)
keys = ("kmer", "means", "stds", "sanums", "signals")
d = {k: torch.from_numpy(f[k]).to(dev) for k in keys}
out = {"batch": B, "config": "configs[2]: bf16 conv (+ folded fp32 joint model), fp32 BiLSTM, batch %d" % B}
acts = {}
for prec in ("fp32", "bf16", "bf16_all"):
    e = Engine(device=0, max_batch=B, precision=prec); e.load_weights(w)
    act = torch.zeros((B, 2), dtype=torch.float32, device=dev); pred = torch.zeros((B,), dtype=torch.int32, device=dev)
    def step():
        e.run_device(B, *(d[k].data_ptr() for k in keys), act.data_ptr(), pred.data_ptr())
    K = max(8, 40960 // B)
    for _ in range(3): step()
    e.sync(); t0 = time.perf_counter()
    for _ in range(K): step()
    e.sync(); dt = time.perf_counter() - t0
    acts[prec] = (act.cpu().numpy(), pred.cpu().numpy())
    r = {"ms_per_step": round(1e3 * dt / K, 4), "sites_per_s": round(K * B / dt, 1), "steps": K,
         "tflops_executed": round(K * B / dt * sum(st["flops_per_site"] for st in e.stage_times()) / 1e12, 2),
         "joint_model": "folded (W1 W2 -> 6032 x 2 in float64 at load%s)" % ("; avgpool folded too" if prec == "fp32" else "; bf16 joint row, fp32 matrix")}
    e.set_profiling(1); e.reset_stage_times()
    for _ in range(3): step()
    e.sync()
    ks = {}
    for k in e.kernel_stats():
        if k["launches"]:
            ks[k["name"]] = {"launches_per_step": k["launches"] // 3, "us_per_step": round(1e3 * k["total_ms"] / 3, 1),
                             "tflops": round(k["flops"] / (k["total_ms"] * 1e-3) / 1e12, 2) if k["total_ms"] > 0 else 0.0}
    r["kernels"] = ks
    e.set_profiling(3); e.reset_stage_times()
    for _ in range(3): step()
    e.sync()
    for k in e.kernel_stats():
        if k["launches"] and k["name"] in ks:
            ks[k["name"]]["us_per_step_alone"] = round(1e3 * k["total_ms"] / 3, 1)
    out[prec] = r
    e.close()
a32, p32 = acts["fp32"]
pn = lambda a: a / a.sum(axis=1, keepdims=True)
decided = np.abs(a32[:, 1] - a32[:, 0]) > 2e-2
for prec in ("bf16", "bf16_all"):
    a16, p16 = acts[prec]
    out[prec]["tolerance_vs_fp32"] = {
        "max_abs_d_act": float(np.abs(a16 - a32).max()), "mean_abs_d_act": float(np.abs(a16 - a32).mean()),
        "max_abs_d_pnorm": float(np.abs(pn(a16) - pn(a32)).max()), "mean_abs_d_pnorm": float(np.abs(pn(a16) - pn(a32)).mean()),
        "label_flip_rate": float((p16 != p32).mean()),
        "label_agreement_margin_gt_2e-2": float((p16[decided] == p32[decided]).mean()) if decided.any() else None,
        "decided_sites": int(decided.sum())}
    out[prec]["speedup_vs_fp32"] = round(out[prec]["sites_per_s"] / out["fp32"]["sites_per_s"], 3)
s = json.dumps(out)
print(s)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(s + "\n")
