"""CPU, world_size 2 (gloo): sharding by read + the result gather — the N>1 path of bench.py / the
CLI. Each rank runs the forward of ITS reads (through the CPU oracle here, standing in for the
engine) and rank 0 must reassemble exactly the single-process result."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from deepsignal_amd import sharding, synth, weights
    from oracle import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 23
    feats = synth.synthetic_features(n, seed=99)
    reads = ["read%d" % (i // 4) for i in range(n)]             # 4 sites per read, last read ragged
    w = weights.random_weights(seed=11, lstm_bias_std=0.1)
    mine = sharding.shard_indices(reads, world, rank)
    sub = {k: v[mine] for k, v in feats.items()}
    act, pred = oracle.forward(w, sub, "f32", nthreads=2) if len(mine) else (np.zeros((0, 2), np.float32), np.zeros((0,), np.int32))
    g_act, g_pred = sharding.gather_results(act, pred, mine, dist)
    t_act, t_pred = sharding.gather_results(act, pred, mine, dist, as_numpy=False)      # tensor form used by bench.py
    if rank == 0:
        assert np.array_equal(t_act.numpy(), g_act) and np.array_equal(t_pred.numpy(), g_pred)
        np.savez(os.path.join(tmp, "gathered.npz"), act=g_act, pred=g_pred)
    else:
        assert g_act is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_by_read_is_a_partition():
    from deepsignal_amd import sharding
    reads = ["a", "a", "b", "c", "c", "c", "a", "d"]      # 'a' re-appears: still one owner
    ranks = sharding.assign_reads(reads, 3)
    assert list(ranks) == [0, 0, 1, 2, 2, 2, 0, 0]
    parts = [sharding.shard_indices(reads, 3, r) for r in range(3)]
    assert sorted(np.concatenate(parts).tolist()) == list(range(len(reads)))


def test_two_rank_gather_equals_single_process(tmp_path):
    sys.path.insert(0, ROOT)
    from deepsignal_amd import synth, weights
    from oracle import oracle
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "gathered.npz"))
    feats = synth.synthetic_features(23, seed=99)
    w = weights.random_weights(seed=11, lstm_bias_std=0.1)
    act, pred = oracle.forward(w, feats, "f32", nthreads=2)
    assert np.array_equal(got["act"], act) and np.array_equal(got["pred"], pred)


# ---- the harness-level N>1 path: `call_mods` launched one process per GPU (here: 2 gloo ranks, oracle engines) ----
def _write_feature_tsv(path, feats, reads):
    bases = "ACGTN"
    with open(path, "w") as f:
        for i in range(len(reads)):
            cols = ["chr1", str(100 + i), "+", str(i), reads[i], "t",
                    "".join(bases[int(c)] for c in feats["kmer"][i]),
                    ",".join("%s" % np.float32(x) for x in feats["means"][i]),
                    ",".join("%s" % np.float32(x) for x in feats["stds"][i]),
                    ",".join(str(int(x)) for x in feats["sanums"][i]),
                    ",".join("%s" % np.float32(x) for x in feats["signals"][i]),
                    str(int(feats["labels"][i]))]
            f.write("\t".join(cols) + "\n")


class _OracleEngine:
    """Stands in for the HIP engine on the CPU-only test box (tests may use the oracle; the product never does)."""
    class_num = 2

    def __init__(self, weights, log):
        self.w, self.log = weights, log

    def run(self, kmer, means, stds, sanums, signals):
        from oracle import oracle
        self.log.append(len(kmer))
        feats = {"kmer": np.asarray(kmer, np.int32), "means": np.asarray(means, np.float32),
                 "stds": np.asarray(stds, np.float32), "sanums": np.asarray(sanums, np.float32),
                 "signals": np.asarray(signals, np.float32)}
        return oracle.forward(self.w, feats, "f32", nthreads=2)


def _call_mods_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from deepsignal_amd import call_modifications as cm, weights
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = weights.random_weights(seed=11, lstm_bias_std=0.1)
    log = []
    n = cm.call_mods(os.path.join(tmp, "features.tsv"), "unused", os.path.join(tmp, "sharded.tsv"), 17, 360, 8, 0.001, 2,
                     1, True, True, True, True, (2,), engine=_OracleEngine(w, log), dist=dist)
    with open(os.path.join(tmp, "rank%d.log" % rank), "w") as f:
        f.write("%d %d\n" % (n, sum(log)))
    dist.barrier()
    dist.destroy_process_group()


def test_call_mods_two_ranks_writes_the_single_process_file(tmp_path):
    """5 queue items (2 reads each, ragged last one) over 2 ranks: rank 0's file must equal the single-process file
    byte for byte, and every rank must have run only its own items."""
    sys.path.insert(0, ROOT)
    lib = os.path.join(ROOT, "deepsignal_amd", "libdeepsignal_hip.so")
    if not os.path.exists(lib):
        import pytest
        pytest.skip("native library not built (the sharded harness uses the native reader)")
    from deepsignal_amd import call_modifications as cm, synth, weights
    n = 27
    feats = synth.synthetic_features(n, seed=5)
    reads = ["read%d" % (i // 3) for i in range(n)]               # 9 reads, 3 sites each -> items of 2 reads: 6,6,6,6,3
    tmp = str(tmp_path)
    _write_feature_tsv(os.path.join(tmp, "features.tsv"), feats, reads)
    mp.spawn(_call_mods_worker, args=(2, _free_port(), tmp), nprocs=2, join=True)
    w = weights.random_weights(seed=11, lstm_bias_std=0.1)
    log = []
    cm.call_mods(os.path.join(tmp, "features.tsv"), "unused", os.path.join(tmp, "single.tsv"), 17, 360, 8, 0.001, 2, 1,
                 True, True, True, True, (2,), engine=_OracleEngine(w, log))
    assert open(os.path.join(tmp, "sharded.tsv"), "rb").read() == open(os.path.join(tmp, "single.tsv"), "rb").read()
    r0 = [int(x) for x in open(os.path.join(tmp, "rank0.log")).read().split()]
    r1 = [int(x) for x in open(os.path.join(tmp, "rank1.log")).read().split()]
    assert r0[0] == r1[0] == n                   # both saw all the rows go by
    assert r0[1] == 6 + 6 + 3 and r1[1] == 6 + 6  # items 0,2,4 on rank 0; items 1,3 on rank 1
    assert not os.path.exists(os.path.join(tmp, "sharded.tsv.rank1"))
