#!/usr/bin/env python3
"""bench.py — CpG sites/sec of the call_mods forward on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (ds_forward_device) over one batch of 512 synthetic
(kmer=17, signal=360) sites whose features are already resident in HBM.  N>1: one process per GPU
(torch.distributed / RCCL), sites sharded by read with a full weight replica per rank (weak
scaling, no data-path collective); the only collective is the final result gather to rank 0.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      the dominant kernel's achieved TFLOP/s (algorithmic 2*M*N*K per launch / HIP-event
                duration per launch) against the fp32 MFMA peak
  cpu_baseline  the CPU oracle (oracle/ds_oracle.c, "port") timed on this box's host cores
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 512
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
WORKLOAD = "configs[1]: 1xMI355X, batch=512, random-init CpG model weights, synthetic (17,360) features, fp32"


def effective_cores():
    """Host cores this process may actually use: the GPU boxes expose 256 logical CPUs but run the job under a cgroup
    CPU quota (cpu.max), and an OpenMP / MKL pool wider than the quota thrashes (measured: 256 threads are 2.6x
    slower than 32 for the C port, and PyTorch-CPU drops from 1,400 to 1.5 sites/s)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(-(-int(quota) // int(period)))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(weights, budget_s=12.0):
    """Bounded samples of the same workload through the two CPU stand-ins for the TF1-CPU path, both on
    all usable host cores: the C port (oracle/ds_oracle.c, OpenMP over sites) and the same graph in PyTorch-CPU
    library ops (oracle/torch_statement.py; oneDNN / MKL kernels, the closest analogue of TF1's Eigen / MKL ones).
    The reported baseline is the FASTER of the two; the other one rides along in `other_port`."""
    from deepsignal_amd import synth
    from oracle import oracle
    cores = effective_cores()
    nmax = 16384
    feats = synth.synthetic_features(nmax, seed=synth.FEATURE_SEED)
    nprobe = min(nmax, max(256, 8 * cores))
    probe = {k: v[:nprobe] for k, v in feats.items()}
    oracle.forward(weights, probe, "f32", nthreads=cores)          # warm-up (page-in, thread pool)
    t0 = time.perf_counter()
    oracle.forward(weights, probe, "f32", nthreads=cores)
    rate = nprobe / (time.perf_counter() - t0)
    n = int(min(nmax, max(nprobe, (rate * budget_s) // 512 * 512)))
    sample = {k: v[:n] for k, v in feats.items()}
    t0 = time.perf_counter()
    oracle.forward(weights, sample, "f32", nthreads=cores)
    dt = time.perf_counter() - t0
    c_port = {"value": round(n / dt, 2), "unit": "sites/s", "cores": cores, "kind": "port",
              "sample": "%d synthetic sites of the same workload (batches of 512) through oracle/ds_oracle.c "
                        "(f32, OpenMP, %d threads), %.1f s" % (n, cores, dt)}
    t_port = None
    try:
        import torch
        from oracle import torch_statement
        torch.set_num_threads(cores)
        wt = {k: torch.from_numpy(v) for k, v in weights.items()}
        t0 = time.perf_counter()
        torch_statement.forward(wt, {k: v[:64] for k, v in feats.items()}, dtype=torch.float32)    # warm-up + guard
        if time.perf_counter() - t0 > 20.0:
            raise RuntimeError("PyTorch-CPU probe of 64 sites took %.0f s; leg skipped" % (time.perf_counter() - t0))
        t0 = time.perf_counter()
        torch_statement.forward(wt, {k: v[:512] for k, v in feats.items()}, dtype=torch.float32)
        per = time.perf_counter() - t0
        reps = int(max(1, min(32, budget_s // max(per, 1e-3))))
        t0 = time.perf_counter()
        for r in range(reps):
            torch_statement.forward(wt, {k: v[512 * r:512 * (r + 1)] for k, v in feats.items()}, dtype=torch.float32)
        dt2 = time.perf_counter() - t0
        t_port = {"value": round(512 * reps / dt2, 2), "unit": "sites/s", "cores": cores, "kind": "port",
                  "sample": "%d synthetic sites of the same workload (batches of 512) through oracle/torch_statement.py "
                            "(PyTorch-CPU f32, %d threads), %.1f s" % (512 * reps, cores, dt2)}
    except Exception as exc:      # the baseline leg must not take the bench line down
        c_port["torch_cpu_error"] = repr(exc)
    if t_port is not None:
        best, other = (t_port, c_port) if t_port["value"] > c_port["value"] else (c_port, t_port)
        best["other_port"] = {"value": other["value"], "sample": other["sample"]}
        return best
    return c_port


def pmc_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are
    collected in separate runs of this same command and corrected as MI355X_MICROARCH.md prescribes; see
    tools/pmc_traffic.py). Counters cannot be read from inside the process, so this is the recorded figure."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    norm = lambda s: s.replace(" ", "").replace(",false>", ">").replace(",true>", ",bf16>")
    try:
        rec = json.load(open(path))
        for name, v in rec["kernels"].items():
            if norm(name) == norm(kernel_name):
                return {"traffic": round(v["hbm_bytes_per_launch"]), "traffic_unit": "B/launch",
                        "traffic_fetch": round(v["fetch_bytes_per_launch"]), "traffic_write": round(v["write_bytes_per_launch"]),
                        "traffic_source": "profiles/r01_pmc_traffic.json (" + rec["source"] + ")"}
    except (OSError, ValueError, KeyError):
        pass
    return {"traffic": None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile-pass", action="store_true")
    ap.add_argument("--no-standalone-pass", action="store_true",
                    help="skip pass C (every launch on one stream); used for the rocprofv3 cross-check, whose per-kernel "
                         "averages would otherwise blend co-resident and stand-alone launches")
    args = ap.parse_args()

    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..."
                             % (args.gpus, args.gpus))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from deepsignal_amd import spec, synth, weights as W
    from deepsignal_amd.engine import Engine

    w = W.random_weights(seed=W.WEIGHT_SEED)          # TF-initializer style, randomised BN
    eng = Engine(device=local_rank, max_batch=BATCH)
    eng.load_weights(w)

    # this rank's shard: its own reads (20 sites per read), NPOOL distinct batches resident in HBM
    NPOOL = 8
    feats = synth.synthetic_features(NPOOL * BATCH, seed=synth.FEATURE_SEED + rank)
    d = {k: torch.from_numpy(feats[k]).to(dev) for k in ("kmer", "means", "stds", "sanums", "signals")}
    K, Wm = args.steps, args.warmup
    out_act = torch.zeros((max(K, 1), BATCH, 2), dtype=torch.float32, device=dev)
    out_pred = torch.zeros((max(K, 1), BATCH), dtype=torch.int32, device=dev)

    def step(i, slot):
        b = (i % NPOOL) * BATCH
        eng.run_device(BATCH, d["kmer"][b:b + BATCH].data_ptr(), d["means"][b:b + BATCH].data_ptr(),
                       d["stds"][b:b + BATCH].data_ptr(), d["sanums"][b:b + BATCH].data_ptr(),
                       d["signals"][b:b + BATCH].data_ptr(), out_act[slot].data_ptr(), out_pred[slot].data_ptr())

    def fence():
        eng.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(Wm):
        step(i, 0)
    fence()
    t0 = time.perf_counter()
    for i in range(K):
        step(i, i)
    eng.sync()
    if dist is not None:
        # the path's only exchange: gather f32[n,2] + i32[n] (12 B/site) to the writer rank over RCCL
        from deepsignal_amd import sharding
        gidx = torch.arange(rank, world * K * BATCH, world, dtype=torch.int64, device=dev)   # this rank's global site ids
        g_act, g_pred = sharding.gather_results(out_act.reshape(-1, 2), out_pred.reshape(-1), gidx, dist, dst=0, device=dev,
                                                as_numpy=False)
        if rank == 0:
            assert g_act.shape[0] == world * K * BATCH
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert bool(torch.isfinite(out_act).all())

    result = {
        "metric": "CpG sites/sec (batch=512, k=17, sig=360)",
        "value": round(world * K * BATCH / elapsed, 1) if K else 0.0,
        "unit": "sites/s", "n_gpus": world, "steps": K, "warmup": Wm,
        "ms_per_step": round(1e3 * elapsed / max(K, 1), 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": WORKLOAD, "batch": BATCH, "kmer_len": 17, "signal_len": 360,
                   "sharding": "by read, %d rank(s), full weight replica per GPU" % world},
    }

    if rank == 0 and not args.no_profile_pass:
        # HIP events on the engine's own streams, eager replay of the same K steps.
        # pass A (mode 2, one pair per launch): per-stage breakdown
        KP = min(K, 50)                   # replayed steps per event pass
        eng.set_profiling(2)
        eng.reset_stage_times()
        for i in range(KP):
            step(i, i)
        eng.sync()
        stages = {s["name"]: round(1e3 * s["total_ms"] / max(s["calls"], 1), 1) for s in eng.stage_times()}
        # pass B (mode 1, one pair per run of same-kernel launches): per-kernel durations for the roofline
        eng.set_profiling(1)
        eng.reset_stage_times()
        tp = time.perf_counter()
        for i in range(KP):
            step(i, i)
        eng.sync()
        prof_ms = 1e3 * (time.perf_counter() - tp) / max(KP, 1)
        ks = [k for k in eng.kernel_stats() if k["launches"]]
        # pass C (mode 3): the same, with every launch on one stream -> stand-alone duration of each kernel
        alone = {}
        if not args.no_standalone_pass:
            eng.set_profiling(3)
            eng.reset_stage_times()
            for i in range(KP):
                step(i, i)
            eng.sync()
            alone = {k["name"]: k for k in eng.kernel_stats() if k["launches"]}
        eng.set_profiling(0)
        dom = max(ks, key=lambda k: k["total_ms"])
        per_launch_flops = dom["flops"] / dom["launches"]
        avg_ms = dom["total_ms"] / dom["launches"]
        achieved = per_launch_flops / (avg_ms * 1e-3) / 1e12
        result["roofline"] = {
            "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None,
            "kernel": dom["name"], "launches_per_step": dom["launches"] // max(KP, 1),
            "avg_launch_us": round(avg_ms * 1e3, 2), "flops_per_launch": per_launch_flops,
            "profiled_ms_per_step": round(prof_ms, 4),
            "whole_path_tflops": round(spec.FLOPS_PER_SITE * result["value"] / world / 1e12, 2),
        }
        if dom["name"] in alone:     # the same kernel with the GPU to itself (no signal-model kernels co-resident)
            a = alone[dom["name"]]
            a_ms = a["total_ms"] / a["launches"]
            a_tf = per_launch_flops / (a_ms * 1e-3) / 1e12
            result["roofline"]["standalone"] = {"avg_launch_us": round(a_ms * 1e3, 2), "achieved": round(a_tf, 2),
                                                "frac": round(a_tf / PEAK_FP32_MFMA_TFLOPS, 4)}
        result["roofline"].update(pmc_traffic(dom["name"]))
        result["kernels"] = {k["name"]: {"launches_per_step": k["launches"] // max(KP, 1),
                                          "us_per_step": round(1e3 * k["total_ms"] / max(KP, 1), 1),
                                          "tflops": round(k["flops"] / (k["total_ms"] * 1e-3) / 1e12, 2) if k["total_ms"] else 0,
                                          "us_per_step_alone": round(1e3 * alone[k["name"]]["total_ms"] / max(KP, 1), 1)
                                          if k["name"] in alone else None}
                             for k in ks}
        result["stages_us_per_step"] = stages
    eng.close()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(w)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
