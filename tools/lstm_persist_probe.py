"""DS_LSTM_TILING_PERSISTENT (the bf16-operand BiLSTM of a forward as ONE persistent launch) against the diagonal launches:
bit-equality of the outputs and of every LSTM tap, stand-alone kernel time, step rate.   usage: lstm_persist_probe.py [batch] [steps]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from deepsignal_amd import synth, weights as W
from deepsignal_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
w = W.random_weights(seed=7, lstm_bias_std=0.1)
keys = ("kmer", "means", "stds", "sanums", "signals")
f = synth.synthetic_features(B, seed=3)
args = [f[k] for k in keys]
out = {}
for tag, kw in (("diagonal", {}), ("persistent", {"lstm_tiling": "persistent"})):
    # taps (debug engines keep every H buffer) on a ragged batch
    e = Engine(max_batch=B, precision="bf16_all", debug=True, slots=1, **kw); e.load_weights(w)
    m = min(B, 2100)
    a, p = e.run(*(x[:m] for x in args))
    taps = {n_: e.intermediate(n_, (m, 17, 256)) for n_ in ("lstm_fw_l0", "lstm_bw_l0", "lstm_fw_l1", "lstm_bw_l1", "lstm_fw_l2", "lstm_bw_l2")}
    e.close()
    e = Engine(max_batch=B, precision="bf16_all", **kw); e.load_weights(w)
    a_full, p_full = e.run(*args)
    a_small, _ = e.run(*(x[100:177] for x in args))
    import torch
    dev = torch.device("cuda", 0)
    d = {k: torch.from_numpy(f[k]).to(dev) for k in keys}
    act = torch.zeros((B, 2), dtype=torch.float32, device=dev); pred = torch.zeros((B,), dtype=torch.int32, device=dev)
    step = lambda: e.run_device(B, *(d[k].data_ptr() for k in keys), act.data_ptr(), pred.data_ptr())
    for _ in range(6): step()
    e.sync()
    rates = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps): step()
        e.sync()
        rates.append(steps * B / (time.perf_counter() - t0))
    print("   device-resident run == host run:", np.array_equal(act.cpu().numpy(), a_full))
    e.set_profiling(3); e.reset_stage_times()
    for _ in range(6): step()
    e.sync()
    ks = {k["name"]: round(1e3 * k["total_ms"] / 6, 1) for k in e.kernel_stats() if k["launches"]}
    e.set_profiling(0)
    e.close()
    out[tag] = (a, p, taps, a_full, a_small)
    print("%-10s %.0f sites/s (%s); stand-alone us per step: %s" % (tag, sorted(rates)[1], " ".join("%.0f" % r for r in rates), ks), flush=True)
a0, p0, t0_, f0, s0 = out["diagonal"]; a1, p1, t1_, f1, s1 = out["persistent"]
for k in t0_:
    print("%-12s equal bits: %s   max|d| %.3g" % (k, np.array_equal(t0_[k], t1_[k]), float(np.abs(t0_[k] - t1_[k]).max())))
print("outputs (2100, debug) equal:", np.array_equal(a0, a1), " full batch equal:", np.array_equal(f0, f1), " sub-batch equal:", np.array_equal(s0, s1),
      " sub-batch == rows of the full batch:", np.array_equal(s1, f1[100:177]))
