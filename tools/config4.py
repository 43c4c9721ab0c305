"""BASELINE configs[3]: per-read shard of 10 M synthetic sites over the GPUs of one node, RCCL gather of the results.

    python tools/config4.py [--sites 10000000] [--batch 512] [--precision fp32]                      # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/config4.py   # 8 GPUs

20 sites per read, reads dealt round-robin to the ranks (deepsignal_amd.sharding), every rank runs its shard through
ds_forward_device in `batch`-site forwards from a resident pool of synthetic batches (the shard is far larger than the
pool; features repeat, work does not), results stay on the device, and the run ends with ONE gather of
f32[n_i,2] + i32[n_i] to rank 0. Strong scaling: total work is fixed. Prints one JSON line on rank 0.

`run_shard()` is the same path as a function (tests/test_gpu_configs.py drives it at >= 1 M sites on one GPU and checks
sampled sites against the oracle)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

KEYS = ("kmer", "means", "stds", "sanums", "signals")
SITES_PER_READ = 20


def pool_features(pool_sites, rank=0):
    from deepsignal_amd import synth
    return synth.synthetic_features(pool_sites, seed=synth.FEATURE_SEED + rank)


def gpu_telemetry():
    """Clocks / temperature / power of GPU 0 as rocm-smi reports them (an ordinary user may read them): sampled before, in the
    middle of and after the sustained run, so that a reader can tell a throttled run from a quiet one."""
    import subprocess
    try:
        out = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showtemp", "--showpower", "--json"], capture_output=True,
                             text=True, timeout=20).stdout
        card = next(iter(json.loads(out).values()))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(w in kl for w in ("sclk", "mclk", "fclk", "temperature", "power")):
                keep[k] = v
        return keep
    except Exception as exc:       # rocm-smi absent or unreadable: the run itself does not depend on it
        return {"unavailable": str(exc)[:80]}


def run_shard(sites, batch=512, precision="fp32", weights=None, pool_sites=4096, dist=None, rank=0, local=0, world=1, telemetry=False):
    """This rank's reads of a `sites`-site job through the resident-input boundary, then the one result gather.
    Returns (record, gathered act, gathered pred, my_reads); the gathered tensors are on rank 0's device (None elsewhere)."""
    import torch
    from deepsignal_amd import sharding, weights as W
    from deepsignal_amd.engine import Engine
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    B = batch
    # shard by read: read r (sites 20r .. 20r+19) belongs to rank r % world
    nreads = sites // SITES_PER_READ
    my_reads = np.arange(rank, nreads, world, dtype=np.int64)
    n_mine = int(my_reads.size) * SITES_PER_READ
    eng = Engine(device=local, max_batch=B, precision=precision)
    eng.load_weights(weights if weights is not None else W.random_weights(seed=W.WEIGHT_SEED))
    NPOOL = max(1, pool_sites // B)
    f = pool_features(NPOOL * B, rank)
    d = {k: torch.from_numpy(f[k]).to(dev) for k in KEYS}
    nsteps = (n_mine + B - 1) // B
    out_act = torch.zeros((nsteps * B, 2), dtype=torch.float32, device=dev)
    out_pred = torch.zeros((nsteps * B,), dtype=torch.int32, device=dev)

    def step(i):
        b = (i % NPOOL) * B
        eng.run_device(B, *(d[k][b:b + B].data_ptr() for k in KEYS), out_act[i * B:].data_ptr(), out_pred[i * B:].data_ptr())

    for i in range(min(8, nsteps)):
        step(i)
    eng.sync(); torch.cuda.synchronize()
    if dist is not None:
        dist.barrier(); torch.cuda.synchronize()
    tele = [gpu_telemetry()] if telemetry else None
    mid = {}
    sampler = None
    if telemetry:
        # the mid-run sample forks rocm-smi (a Python program: hundreds of ms). It runs on a helper thread that the timed loop
        # only signals, so the feeding thread never waits for it; how long the sample took is recorded
        import threading
        go = threading.Event()

        def _sample():
            go.wait()
            ts = time.perf_counter()
            mid["sample"] = gpu_telemetry()
            mid["seconds"] = round(time.perf_counter() - ts, 3)
        sampler = threading.Thread(target=_sample, daemon=True)
        sampler.start()
    t0 = time.perf_counter()
    for i in range(nsteps):
        step(i)
        if telemetry and i == nsteps // 2:
            go.set()
    eng.sync()
    t_compute = time.perf_counter() - t0
    if telemetry:
        go.set()
        sampler.join()
        tele.append(mid.get("sample"))
        tele.append(gpu_telemetry())
    # global site index of my j-th site: read my_reads[j // 20], position j % 20
    def index_of_rank(r, cnt):       # the sharding rule: rank r owns reads r, r + world, ...; 20 sites per read
        reads = torch.arange(r, r + world * (cnt // SITES_PER_READ), world, dtype=torch.int64, device=dev)
        return (reads.repeat_interleave(SITES_PER_READ) * SITES_PER_READ + torch.arange(SITES_PER_READ, device=dev).repeat(reads.numel()))
    g_act, g_pred = sharding.gather_results(out_act[:n_mine], out_pred[:n_mine], None, dist, dst=0, device=dev, as_numpy=False,
                                            index_of_rank=index_of_rank, force_collective=dist is not None)      # one rank under the launcher: same RCCL calls
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed, t_compute], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, t_compute = float(t[0]), float(t[1])
    eng.close()
    total = nreads * SITES_PER_READ
    rec = {"config": "configs[3]: per-read shard of %d synthetic sites, %d GPU(s), batch %d, %s" % (total, world, B, precision),
           "n_gpus": world, "sites": total, "seconds": round(elapsed, 3), "seconds_compute_max_rank": round(t_compute, 3),
           "sites_per_s": round(total / elapsed, 1), "sites_per_s_compute_only": round(total / t_compute, 1),
           "seconds_gather": round(elapsed - t_compute, 3), "gather_bytes": total * 12, "pool_sites": NPOOL * B}
    if telemetry:
        rec["gpu_telemetry_before_mid_after"] = tele
        rec["mid_sample_seconds_on_helper_thread"] = mid.get("seconds")
    return rec, g_act, g_pred, my_reads


def main():
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--sites", type=int, default=10_000_000)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--telemetry", action="store_true", help="record rocm-smi clocks / temperature / power before, during and after")
    args = ap.parse_args()
    rank, local, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    rec, g_act, g_pred, _ = run_shard(args.sites, args.batch, args.precision, dist=dist, rank=rank, local=local, world=world,
                                       telemetry=args.telemetry and rank == 0)
    if rank == 0:
        if world > 1:
            assert g_act.shape[0] == rec["sites"] and bool(torch.isfinite(g_act).all())
        print(json.dumps(rec))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
