"""Host-core probe for the cpu_baseline leg: effective cores and oracle / torch-CPU rates vs thread count."""
import os, sys, time, subprocess
sys.path.insert(0, os.getcwd())
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(p, open(p).read().strip())
    except OSError as e: print(p, "n/a")
print(subprocess.run("nproc; lscpu | grep -E 'Model name|Socket|Core|Thread|^CPU\\(s\\)'; cat /proc/loadavg", shell=True, capture_output=True, text=True).stdout)
import numpy as np
from deepsignal_amd import synth, weights as W
from oracle import oracle
w = W.random_weights()
f = synth.synthetic_features(2048, seed=1)
for th in (8, 16, 32, 64, 128, 256):
    n = 1024
    sub = {k: v[:n] for k, v in f.items()}
    oracle.forward(w, {k: v[:256] for k, v in f.items()}, "f32", nthreads=th)
    t0 = time.perf_counter(); oracle.forward(w, sub, "f32", nthreads=th); dt = time.perf_counter() - t0
    print("oracle threads %3d: %.0f sites/s" % (th, n / dt), flush=True)
code = r'''
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
th = int(sys.argv[1]); torch.set_num_threads(th)
from deepsignal_amd import synth, weights as W
from oracle import torch_statement
w = {k: torch.from_numpy(v) for k, v in W.random_weights().items()}
f = synth.synthetic_features(256, seed=1)
torch_statement.forward(w, f, dtype=torch.float32)
t0 = time.perf_counter(); torch_statement.forward(w, f, dtype=torch.float32); dt = time.perf_counter() - t0
print("torch threads %3d: %.0f sites/s" % (th, 256 / dt), flush=True)
'''
for th in (8, 16, 32, 64):
    try:
        r = subprocess.run([sys.executable, "-c", code, str(th)], capture_output=True, text=True, timeout=60)
        print(r.stdout.strip() or r.stderr[-300:])
    except subprocess.TimeoutExpired:
        print("torch threads %d: timeout (60 s)" % th)
