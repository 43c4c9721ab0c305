"""Throughput vs batch size per forward (diagnostic; the headline metric is batch 512)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from deepsignal_amd import synth, weights as W
from deepsignal_amd.engine import Engine
w = W.random_weights(seed=W.WEIGHT_SEED)
dev = torch.device("cuda", 0)
for B in [int(x) for x in sys.argv[1:]] or [128, 256, 512, 1024, 2048, 4096]:
    e = Engine(device=0, max_batch=B); e.load_weights(w)
    f = synth.synthetic_features(B, seed=1)
    d = {k: torch.from_numpy(f[k]).to(dev) for k in ("kmer", "means", "stds", "sanums", "signals")}
    act = torch.zeros((B, 2), dtype=torch.float32, device=dev); pred = torch.zeros((B,), dtype=torch.int32, device=dev)
    K = max(4, 20480 // B)
    def step():
        e.run_device(B, d["kmer"].data_ptr(), d["means"].data_ptr(), d["stds"].data_ptr(), d["sanums"].data_ptr(), d["signals"].data_ptr(), act.data_ptr(), pred.data_ptr())
    for _ in range(3): step()
    e.sync(); t0 = time.perf_counter()
    for _ in range(K): step()
    e.sync(); dt = time.perf_counter() - t0
    flops = sum(st["flops_per_site"] for st in e.stage_times())          # FLOPs the engine executes per site
    print("batch %5d: %.3f ms/step, %.0f sites/s, %.1f TFLOP/s executed" % (B, 1e3 * dt / K, K * B / dt, K * B / dt * flops / 1e12))
    e.close()
