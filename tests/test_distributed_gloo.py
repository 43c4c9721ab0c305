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
    if rank == 0:
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
