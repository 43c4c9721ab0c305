"""bf16x3 in the pipelined step: per-kernel durations while the forwards of all slots share the GPU (profiling mode 1) next to the
stand-alone ones (mode 3). usage: python tools/attic/r05_x3_mix.py [fold=0|1]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from deepsignal_amd import synth, weights as W
from deepsignal_amd.engine import Engine
fold = len(sys.argv) > 1 and sys.argv[1] == "1"
B = 512; dev = torch.device("cuda", 0)
w = W.random_weights(seed=W.WEIGHT_SEED)
f = synth.synthetic_features(8 * B, seed=1)
keys = ("kmer", "means", "stds", "sanums", "signals")
d = {k: torch.from_numpy(f[k]).to(dev) for k in keys}
act = torch.zeros((B, 2), dtype=torch.float32, device=dev); pred = torch.zeros((B,), dtype=torch.int32, device=dev)
for prec in ("bf16x3", "fp32"):
    e = Engine(max_batch=B, precision=prec, fold_fc=fold); e.load_weights(w)
    def step(i):
        b = (i % 8) * B
        e.run_device(B, *(d[k][b:b + B].data_ptr() for k in keys), act.data_ptr(), pred.data_ptr())
    for i in range(20): step(i)
    e.sync()
    t0 = time.perf_counter()
    for i in range(200): step(i)
    e.sync(); us = (time.perf_counter() - t0) / 200 * 1e6
    res = {}
    for mode in (1, 3):
        e.set_profiling(mode); e.reset_stage_times()
        for i in range(40): step(i)
        e.sync()
        res[mode] = {k["name"]: 1e3 * k["total_ms"] / 40 for k in e.kernel_stats() if k["launches"]}
    e.set_profiling(0)
    print("%s fold=%d: %.0f us per pipelined step (%.0f sites/s); sum alone %.0f, sum co-resident %.0f" % (prec, fold, us, B / us * 1e6, sum(res[3].values()), sum(res[1].values())))
    for k in sorted(res[3], key=lambda k: -res[3][k]):
        print("   %-52s alone %7.1f   co-resident %7.1f" % (k[:52], res[3][k], res[1].get(k, 0.0)))
    e.close()
