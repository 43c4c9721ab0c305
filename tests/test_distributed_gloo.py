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
    # 12 B/site form: the writer derives the indices from the sharding rule instead of receiving them
    import torch
    rule = lambda r, cnt: torch.from_numpy(sharding.shard_indices(reads, world, r)[:cnt])
    r_act, r_pred = sharding.gather_results(act, pred, None, dist, index_of_rank=rule)
    if rank == 0:
        assert np.array_equal(t_act.numpy(), g_act) and np.array_equal(t_pred.numpy(), g_pred)
        assert np.array_equal(r_act, g_act) and np.array_equal(r_pred, g_pred)
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


def _call_mods_worker(rank, world, port, tmp, chunk_bytes):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from deepsignal_amd import call_modifications as cm, weights
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cm.SHARD_CHUNK_BYTES = chunk_bytes            # small ranges: several rounds of the row gather even on a test-sized file
    w = weights.random_weights(seed=11, lstm_bias_std=0.1)
    log = []
    n = cm.call_mods(os.path.join(tmp, "features.tsv"), "unused", os.path.join(tmp, "sharded.tsv"), 17, 360, 8, 0.001, 2,
                     1, True, True, True, True, None, engine=_OracleEngine(w, log), dist=dist, f5_batch_num=2)
    with open(os.path.join(tmp, "rank%d.log" % rank), "w") as f:
        f.write("%d %d\n" % (n, sum(log)))
    dist.barrier()
    dist.destroy_process_group()


def _need_native():
    if not os.path.exists(os.path.join(ROOT, "deepsignal_amd", "libdeepsignal_hip.so")):
        import pytest
        pytest.skip("native library not built (the sharded harness uses the native reader)")


def _run_sharded_vs_single(tmp, world, reads, chunk_bytes, seed=5):
    from deepsignal_amd import call_modifications as cm, synth, weights
    n = len(reads)
    feats = synth.synthetic_features(n, seed=seed)
    _write_feature_tsv(os.path.join(tmp, "features.tsv"), feats, reads)
    mp.spawn(_call_mods_worker, args=(world, _free_port(), tmp, chunk_bytes), nprocs=world, join=True)
    w = weights.random_weights(seed=11, lstm_bias_std=0.1)
    log = []
    cm.call_mods(os.path.join(tmp, "features.tsv"), "unused", os.path.join(tmp, "single.tsv"), 17, 360, 8, 0.001, 2, 1,
                 True, True, True, True, None, engine=_OracleEngine(w, log), f5_batch_num=2)
    assert open(os.path.join(tmp, "sharded.tsv"), "rb").read() == open(os.path.join(tmp, "single.tsv"), "rb").read()
    logs = [[int(x) for x in open(os.path.join(tmp, "rank%d.log" % r)).read().split()] for r in range(world)]
    assert all(l[0] == n for l in logs)              # every rank returns the job-wide site count
    assert sum(l[1] for l in logs) == n              # and together they ran every site exactly once
    return [l[1] for l in logs]


def test_call_mods_two_ranks_writes_the_single_process_file(tmp_path):
    """9 reads of 3 sites over 2 ranks, byte ranges cut at read boundaries: rank 0's file must equal the single-process
    file byte for byte, and each rank parses and runs only the reads of its own ranges."""
    sys.path.insert(0, ROOT)
    _need_native()
    reads = ["read%d" % (i // 3) for i in range(27)]
    own = _run_sharded_vs_single(str(tmp_path), 2, reads, chunk_bytes=1 << 30)       # one range per rank
    assert all(x % 3 == 0 and x > 0 for x in own)                                    # whole reads only
    assert not os.path.exists(os.path.join(str(tmp_path), "sharded.tsv.rank1"))


def test_call_mods_four_ranks_uneven_reads_several_rounds(tmp_path):
    """World 4, reads of very different lengths (1 .. 13 sites), ranges of ~12 kB so that every rank owns several and
    the row gather runs several rounds: still the single-process file, byte for byte."""
    sys.path.insert(0, ROOT)
    _need_native()
    lens = [1, 7, 2, 13, 1, 1, 5, 3, 9, 2, 4, 6]
    reads = [("r%02d" % k) for k, m in enumerate(lens) for _ in range(m)]
    own = _run_sharded_vs_single(str(tmp_path), 4, reads, chunk_bytes=12000, seed=6)
    assert sum(own) == sum(lens)


def test_call_mods_four_ranks_with_empty_shards(tmp_path):
    """Two reads for four ranks: at least two ranks own no read at all and still take part in every collective."""
    sys.path.insert(0, ROOT)
    _need_native()
    reads = ["a"] * 5 + ["b"] * 2
    own = _run_sharded_vs_single(str(tmp_path), 4, reads, chunk_bytes=1 << 30, seed=7)
    assert sorted(own) == [0, 0, 2, 5]


def test_cut_points_tile_the_file_in_whole_reads(tmp_path):
    """ds_tsv_align: the cut points every rank computes on its own tile the file exactly, and no read straddles one."""
    sys.path.insert(0, ROOT)
    _need_native()
    from deepsignal_amd import fastio, synth
    lens = [3, 1, 8, 2, 2, 11, 1, 4]
    reads = [("q%d" % k) for k, m in enumerate(lens) for _ in range(m)]
    path = os.path.join(str(tmp_path), "f.tsv")
    _write_feature_tsv(path, synth.synthetic_features(len(reads), seed=8), reads)
    rd = fastio.FeatureReader(path, nthreads=1)
    whole = [it for it in rd.items(1)]
    for nchunks in (1, 2, 3, 5, 16, 64):
        cuts = rd.cut_points(nchunks)
        assert cuts[0] == 0 and cuts[-1] == rd.size and cuts == sorted(cuts)
        got = []
        for c in range(nchunks):
            rd.set_range(cuts[c], cuts[c + 1])
            got.extend(rd.items(1))                  # items of ONE read each
        assert [bytes(g.info) for g in got] == [bytes(w_.info) for w_ in whole]
        assert all(np.array_equal(g.signals, w_.signals) for g, w_ in zip(got, whole))
    rd.close()


# ---- the fast5-directory route under WORLD_SIZE > 1: file batch k belongs to rank k % world ----
def _fast5_worker(rank, world, port, tmp, golden):
    sys.path.insert(0, ROOT)
    import json
    import torch.distributed as dist
    from deepsignal_amd import call_modifications as cm, extract_features as ef
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = json.load(open(golden))

    def fake_read(path, corrected_group, basecall_subgroup):
        r = g["reads"][os.path.basename(path)[:-6]]
        return (np.asarray(r["signal"], np.int16), r["starts"], r["lengths"], r["bases"], r["range"] / r["digitisation"],
                r["offset"], (r["read_id"], r["strand"], r["alignstrand"], r["chrom"], r["chrom_start"]))

    ef._read_fast5 = fake_read
    f5_args = (True, "RawGenomeCorrected_000", "BaseCalled_template", None, True, "mad", "CG", 0, 1, 1, None)
    n = cm.call_mods(os.path.join(tmp, "f5"), "unused", os.path.join(tmp, "sharded.tsv"), 17, 360, 16, 0.001, 2, 1, False,
                     True, True, True, f5_args, engine=_SignalEngine(), dist=dist)
    open(os.path.join(tmp, "f5rank%d.log" % rank), "w").write(str(n))
    dist.barrier()
    dist.destroy_process_group()


class _SignalEngine:
    """Deterministic stand-in: outputs depend on the site's own features only (not on the 360 central samples: one
    golden read takes the reference's random.sample branch, extract_features.py:165-168)."""
    class_num = 2

    def run(self, kmer, means, stds, sanums, signals):
        x = np.stack([np.asarray(means, np.float32)[:, 0], np.asarray(stds, np.float32)[:, 1]], axis=1)
        act = np.stack([1 / (1 + np.exp(-x[:, 0])), 1 / (1 + np.exp(x[:, 1]))], axis=1).astype(np.float32)
        return act, np.argmax(act, axis=1)


def test_fast5_directory_two_ranks_writes_the_single_process_file(tmp_path, monkeypatch):
    """Directory input under torch.distributed: batches of files are dealt to the ranks, rank 0 writes the rows in batch
    order (the single-process file), the other ranks write nothing."""
    sys.path.insert(0, ROOT)
    import json
    from deepsignal_amd import call_modifications as cm, extract_features as ef
    golden = os.path.join(ROOT, "tests", "golden", "extract_golden.json")
    g = json.load(open(golden))
    tmp = str(tmp_path)
    os.mkdir(os.path.join(tmp, "f5"))
    for name in g["read_order"]:
        open(os.path.join(tmp, "f5", name + ".fast5"), "wb").close()
    mp.spawn(_fast5_worker, args=(2, _free_port(), tmp, golden), nprocs=2, join=True)

    def fake_read(path, corrected_group, basecall_subgroup):
        r = g["reads"][os.path.basename(path)[:-6]]
        return (np.asarray(r["signal"], np.int16), r["starts"], r["lengths"], r["bases"], r["range"] / r["digitisation"],
                r["offset"], (r["read_id"], r["strand"], r["alignstrand"], r["chrom"], r["chrom_start"]))

    monkeypatch.setattr(ef, "_read_fast5", fake_read)
    f5_args = (True, "RawGenomeCorrected_000", "BaseCalled_template", None, True, "mad", "CG", 0, 1, 1, None)
    n = cm.call_mods(os.path.join(tmp, "f5"), "unused", os.path.join(tmp, "single.tsv"), 17, 360, 16, 0.001, 2, 1, False,
                     True, True, True, f5_args, engine=_SignalEngine())
    assert n > 0 and open(os.path.join(tmp, "f5rank0.log")).read() == open(os.path.join(tmp, "f5rank1.log")).read() == str(n)
    assert open(os.path.join(tmp, "sharded.tsv"), "rb").read() == open(os.path.join(tmp, "single.tsv"), "rb").read()
