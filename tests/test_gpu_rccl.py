"""GPU: the multi-GPU exchange path on REAL RCCL with the one GPU a test box has (BASELINE configs[3]'s mechanism).

A one-rank `nccl` process group issues the same RCCL calls as eight ranks do, so the first librccl call of this project
does not happen on the driver's 8-GPU box:

  (a) sharding.gather_results with the world-of-one shortcut bypassed (force_collective);
  (b) sharding.OrderedRowGather on cuda:0 -- collectives from its communication thread -- while the main thread keeps
      the engine's 8 pipeline slots (16 HIP streams) busy;
  (c) call_mods through the sharded route end to end (byte ranges -> engine -> formatted rows -> RCCL row gather ->
      rank 0's file), byte-identical to the ordinary single-process file;
  (d) `python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`: the launcher contract with the result
      gather inside the timed window.

The group lives in the pytest process itself (the driver records which native libraries that process loaded: librccl
must be among them); (d) is a child process.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from deepsignal_amd import synth, weights as W

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("kmer", "means", "stds", "sanums", "signals")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def rccl():
    """A one-rank process group on backend "nccl" (= RCCL on ROCm) in this process."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        yield dist
    finally:
        dist.destroy_process_group()


@pytest.fixture(scope="module")
def engine512():
    from deepsignal_amd.engine import Engine
    eng = Engine(device=0, max_batch=512)
    eng.load_weights(W.random_weights(seed=21, lstm_bias_std=0.1))
    yield eng
    eng.close()


def _librccl_mapped():
    with open("/proc/self/maps") as f:
        return any("librccl" in line for line in f)


def test_gather_results_runs_on_rccl_with_one_rank(rccl):
    """(a) all_gather of the counts + the two ragged gathers, on device tensors and on host arrays: the re-ordered
    result equals the input permuted by the index rule, bit for bit."""
    import torch
    from deepsignal_amd import sharding
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    n = 4099
    act = rng.random((n, 2), dtype=np.float32)
    pred = rng.integers(0, 2, n).astype(np.int32)
    perm = rng.permutation(n).astype(np.int64)          # global index of local row i
    want_act = np.empty_like(act); want_act[perm] = act
    want_pred = np.empty_like(pred); want_pred[perm] = pred
    # device tensors, index derived from the rule (12 B/site on the wire) -- what bench.py does
    t_act, t_pred = sharding.gather_results(torch.from_numpy(act).to(dev), torch.from_numpy(pred).to(dev), None, rccl, dst=0,
                                            device=dev, as_numpy=False, force_collective=True,
                                            index_of_rank=lambda r, cnt: torch.from_numpy(perm[:cnt]).to(dev))
    assert t_act.is_cuda and np.array_equal(t_act.cpu().numpy(), want_act) and np.array_equal(t_pred.cpu().numpy(), want_pred)
    # host arrays with an explicit index gather (third collective)
    g_act, g_pred = sharding.gather_results(act, pred, perm, rccl, device=dev, force_collective=True)
    assert np.array_equal(g_act, want_act) and np.array_equal(g_pred, want_pred)
    # empty shard
    e_act, e_pred = sharding.gather_results(act[:0], pred[:0], perm[:0], rccl, device=dev, force_collective=True)
    assert e_act.shape == (0, 2) and e_pred.shape == (0,)
    assert _librccl_mapped(), "backend nccl did not load librccl into this process"


def test_ordered_row_gather_on_cuda_beside_a_busy_engine(rccl, engine512, tmp_path):
    """(b) the communication thread runs RCCL collectives on cuda:0 while the main thread keeps 8 forwards in flight on
    the engine's 16 streams: the written file is the rounds' bytes in order, and the forwards issued meanwhile give the
    bits of a quiet reference pass."""
    import torch
    from deepsignal_amd import sharding
    dev = torch.device("cuda", 0)
    eng = engine512
    B, NB = 512, 4
    feats = synth.synthetic_features(NB * B, seed=31)
    d = {k: torch.from_numpy(feats[k]).to(dev) for k in KEYS}
    ref_act = torch.zeros((NB, B, 2), dtype=torch.float32, device=dev)
    ref_pred = torch.zeros((NB, B), dtype=torch.int32, device=dev)

    def forward(i, act, pred):
        b = (i % NB) * B
        eng.run_device(B, *(d[k][b:b + B].data_ptr() for k in KEYS), act.data_ptr(), pred.data_ptr())

    for i in range(NB):
        forward(i, ref_act[i], ref_pred[i])
    eng.sync()
    torch.cuda.synchronize()

    rounds = 40
    rng = np.random.default_rng(7)
    payload = [bytes(rng.integers(32, 127, int(rng.integers(0, 200000)), dtype=np.uint8)) if r % 7 else b"" for r in range(rounds)]
    out = os.path.join(str(tmp_path), "rows.bin")
    gather = sharding.OrderedRowGather(rccl, 0, 1, out, nrounds=rounds, device=dev, depth=4)
    steps = 25 * rounds
    act = torch.zeros((steps, B, 2), dtype=torch.float32, device=dev)
    pred = torch.zeros((steps, B), dtype=torch.int32, device=dev)
    for r in range(rounds):
        for j in range(25):
            forward(r * 25 + j, act[r * 25 + j], pred[r * 25 + j])
        gather.put(payload[r])
    total, errors = gather.close(nsites=steps * B, nerrors=0)
    eng.sync()
    torch.cuda.synchronize()
    assert (total, errors) == (steps * B, 0)
    assert open(out, "rb").read() == b"".join(payload)
    for i in range(steps):
        assert torch.equal(act[i], ref_act[i % NB]) and torch.equal(pred[i], ref_pred[i % NB])


def _write_feature_tsv(path, feats, reads):
    bases = "ACGTN"
    with open(path, "w") as f:
        for i in range(len(reads)):
            cols = ["chr1", str(100 + i), "+", str(i), reads[i], "t", "".join(bases[int(c)] for c in feats["kmer"][i]),
                    ",".join("%.6f" % x for x in feats["means"][i]), ",".join("%.6f" % x for x in feats["stds"][i]),
                    ",".join(str(int(x)) for x in feats["sanums"][i]), ",".join("%.6f" % x for x in feats["signals"][i]),
                    str(int(feats["labels"][i]))]
            f.write("\t".join(cols) + "\n")


def test_call_mods_sharded_route_over_rccl_equals_the_plain_file(rccl, engine512, tmp_path, monkeypatch):
    """(c) 6,000 sites in reads of 1 .. 29 sites, byte ranges of ~0.5 MB (about 40 rounds of the row gather): rank 0's
    file through the sharded route on RCCL equals the ordinary call_mods file byte for byte."""
    from deepsignal_amd import call_modifications as cm
    rng = np.random.default_rng(11)
    lens = []
    while sum(lens) < 6000:
        lens.append(int(rng.integers(1, 30)))
    reads = ["read_%05d" % k for k, m in enumerate(lens) for _ in range(m)]
    n = len(reads)
    feats = synth.synthetic_features(n, seed=41)
    tsv = os.path.join(str(tmp_path), "features.tsv")
    _write_feature_tsv(tsv, feats, reads)
    plain, sharded = os.path.join(str(tmp_path), "plain.tsv"), os.path.join(str(tmp_path), "sharded.tsv")
    args = (17, 360, 512, 0.001, 2, 1, True, True, True, True, None)
    assert cm.call_mods(tsv, "unused", plain, *args, engine=engine512, f5_batch_num=20) == n
    monkeypatch.setattr(cm, "SHARD_CHUNK_BYTES", 1 << 19)
    assert cm.call_mods(tsv, "unused", sharded, *args, engine=engine512, f5_batch_num=20, dist=rccl, force_sharded=True) == n
    a, b = open(plain, "rb").read(), open(sharded, "rb").read()
    assert a.count(b"\n") == n and a == b


def test_config4_shard_with_the_gather_on_rccl(rccl, small_weights):
    """BASELINE configs[3]'s own path (tools/config4.py::run_shard: per-read shard through ds_forward_device, then ONE gather
    of f32[n,2] + i32[n]) with the gather on RCCL: 204,800 sites, same bits as the run without a process group."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import config4
    sites = 204_800
    rec0, a0, p0, _ = config4.run_shard(sites, batch=512, weights=small_weights, pool_sites=4096)
    rec1, a1, p1, _ = config4.run_shard(sites, batch=512, weights=small_weights, pool_sites=4096, dist=rccl, rank=0, local=0, world=1)
    to_np = lambda x: x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)
    assert rec1["sites"] == rec0["sites"] == sites and rec1["gather_bytes"] == sites * 12
    assert hasattr(a1, "is_cuda") and a1.is_cuda                      # the collective path keeps the gathered results on the device
    assert np.array_equal(to_np(a0), to_np(a1)) and np.array_equal(to_np(p0), to_np(p1))


def test_bench_under_the_launcher_with_one_rank(tmp_path):
    """(d) the driver's N > 1 command line with N = 1: process group on RCCL, result gather inside every timed window,
    ONE JSON line with the contract keys."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
           "--windows", "3", "--no-profile-pass", "--sharded-rows-per-rank", "4000"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout.decode()[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["steps"] == 20 and r["value"] > 1e5
    assert r["gather"]["backend"] == "nccl" and r["gather"]["bytes_per_window"] == 20 * 512 * 12
    # the product's sharded TSV -> TSV route inside the same run, its collectives on RCCL (one rank: the calls eight ranks make)
    sh = r["e2e_tsv_sharded"]
    assert "error" not in sh, sh
    assert sh["complete"] and sh["rows"] == 4000 and sh["value"] > 0 and "nccl" in sh["path"]
    with open(os.path.join(str(tmp_path), "bench_launcher_1rank.json"), "w") as f:
        f.write(lines[0])
