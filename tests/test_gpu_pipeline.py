"""GPU: BASELINE.json configs[0] — the whole `deepsignal call_mods` plumbing on 1,000 pre-extracted
synthetic feature rows (k=17, sig=360, 20 rows x 50 reads), batch_size 32: feature TSV -> reader ->
batcher -> HIP engine (weights from a DSAMDW01 file) -> writer, diffed numerically against the same
harness driven by the CPU oracle. Row order, grouping by read and the 10-column contract are exact;
probabilities within 1e-5 (north-star gate 1e-4)."""
import numpy as np
import pytest

from deepsignal_amd import synth, weights as W
from deepsignal_amd.utils.process_utils import code2base_dna

pytestmark = pytest.mark.gpu


def _write_feature_tsv(path, feats, reads):
    with open(path, "w") as f:
        for i in range(len(reads)):
            kmer = "".join(code2base_dna[int(c)] for c in feats["kmer"][i])
            cols = ["chr%d" % (1 + i % 5), str(100 + i), "+-"[i % 2], str(9000 - i), reads[i], "tc"[i % 2], kmer,
                    ",".join("%s" % np.float32(x) for x in feats["means"][i]),
                    ",".join("%s" % np.float32(x) for x in feats["stds"][i]),
                    ",".join(str(int(x)) for x in feats["sanums"][i]),
                    ",".join("%s" % np.float32(x) for x in feats["signals"][i]),
                    str(int(feats["labels"][i]))]
            f.write("\t".join(cols) + "\n")


class OracleEngine:
    def __init__(self, weights):
        self.w = weights

    def run(self, kmer, means, stds, sanums, signals):
        from oracle import oracle
        feats = {"kmer": np.asarray(kmer, np.int32), "means": np.asarray(means, np.float32),
                 "stds": np.asarray(stds, np.float32), "sanums": np.asarray(sanums, np.float32),
                 "signals": np.asarray(signals, np.float32)}
        return oracle.forward(self.w, feats, "f32")


def test_call_mods_cli_1k_rows_batch32(small_weights, tmp_path):
    from deepsignal_amd import call_modifications as cm
    from deepsignal_amd.deepsignal import main
    n = 1000
    feats = synth.synthetic_features(n, seed=321)
    reads = ["read_%04d" % (i // 20) for i in range(n)]
    tsv, wfile = str(tmp_path / "features.tsv"), str(tmp_path / "model.dsw")
    out_gpu, out_cpu = str(tmp_path / "gpu.tsv"), str(tmp_path / "cpu.tsv")
    _write_feature_tsv(tsv, feats, reads)
    W.save_weights(wfile, small_weights)
    assert main(["call_mods", "-i", tsv, "-m", wfile, "-o", out_gpu, "-b", "32", "--nproc", "1", "--is_gpu", "yes"]) == 0
    cm.call_mods(tsv, wfile, out_cpu, 17, 360, 32, 0.001, 2, 1, False, True, True, True, None,
                 engine=OracleEngine(small_weights), f5_batch_num=50)
    g = [l.rstrip("\n").split("\t") for l in open(out_gpu)]
    c = [l.rstrip("\n").split("\t") for l in open(out_cpu)]
    assert len(g) == len(c) == n
    for rg, rc in zip(g, c):
        assert len(rg) == 10 and rg[:6] == rc[:6] and rg[9] == rc[9]          # sample info + k-mer text, same order
        p0, p1 = float(rg[6]), float(rg[7])
        assert abs(p0 - float(rc[6])) <= 1e-5 and abs(p1 - float(rc[7])) <= 1e-5
        assert abs(p0 + p1 - 1.0) <= 1e-6
        if abs(float(rc[7]) - float(rc[6])) > 1e-3:
            assert rg[8] == rc[8]
    assert [r[4] for r in g] == reads                                          # reads stay contiguous and ordered


def test_plain_c_client_matches_python_engine(small_weights, tmp_path):
    """tests/abi_client.c (C99, links only libdeepsignal_hip.so) must produce the Python binding's bits, through the
    blocking call and through ds_submit / ds_wait."""
    import os
    import shutil
    import subprocess
    from deepsignal_amd.engine import Engine, LIB_PATH
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "abi_client")
    subprocess.run([gcc, "-std=c99", "-O1", os.path.join(root, "tests", "abi_client.c"), "-o", exe,
                    "-L" + os.path.dirname(LIB_PATH), "-ldeepsignal_hip", "-Wl,-rpath," + os.path.dirname(LIB_PATH)], check=True)
    n = 150
    feats = synth.synthetic_features(n, seed=99)
    wfile, ffile, ofile = str(tmp_path / "m.dsw"), str(tmp_path / "f.bin"), str(tmp_path / "o.bin")
    W.save_weights(wfile, small_weights)
    with open(ffile, "wb") as f:
        for k, dt in (("kmer", np.int32), ("means", np.float32), ("stds", np.float32), ("sanums", np.float32), ("signals", np.float32)):
            f.write(np.ascontiguousarray(feats[k], dtype=dt).tobytes())
    r = subprocess.run([exe, wfile, ffile, str(n), ofile], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(ofile, dtype=np.uint8)
    act = raw[:n * 8].view(np.float32).reshape(n, 2)
    pred = raw[n * 8:n * 12].view(np.int32)
    act2 = raw[n * 12:n * 20].view(np.float32).reshape(n, 2)
    pred2 = raw[n * 20:].view(np.int32)
    eng = Engine(max_batch=64)
    eng.load_weights(small_weights)
    e_act, e_pred = eng.run(*(feats[k] for k in ("kmer", "means", "stds", "sanums", "signals")))
    eng.close()
    assert np.array_equal(act, e_act) and np.array_equal(pred, e_pred)
    assert np.array_equal(act2, e_act) and np.array_equal(pred2, e_pred)


def test_cli_accepts_a_tf_checkpoint_prefix_and_precision(small_weights, tmp_path, capsys):
    """`--model_path` is a TensorFlow checkpoint prefix in the reference (call_modifications.py:210-211): the CLI must
    take one as is (plus Adam slots it has to ignore) and give the bytes it gives for the flat weight file; and
    `--precision bf16_all` must stay within the documented distance of fp32 and must point its user at `--precision bf16x3`, the
    fast mode with fp32-class results."""
    from deepsignal_amd import tf_checkpoint
    from deepsignal_amd.deepsignal import main
    n = 400
    feats = synth.synthetic_features(n, seed=11)
    reads = ["read_%04d" % (i // 20) for i in range(n)]
    tsv = str(tmp_path / "features.tsv")
    _write_feature_tsv(tsv, feats, reads)
    dsw, ckpt = str(tmp_path / "model.dsw"), str(tmp_path / "bn_17.sn_360.epoch_7.ckpt")
    W.save_weights(dsw, small_weights)
    tensors = dict(small_weights)
    tensors["dense/kernel/Adam"] = np.zeros_like(small_weights["dense/kernel"])
    tensors["beta1_power"] = np.array(0.9, np.float32)
    tf_checkpoint.write_checkpoint(ckpt, tensors)
    outs = {}
    for tag, model, extra in (("dsw", dsw, []), ("ckpt", ckpt, []), ("bf16", ckpt, ["--precision", "bf16_all"])):
        outs[tag] = str(tmp_path / (tag + ".tsv"))
        assert main(["call_mods", "-i", tsv, "-m", model, "-o", outs[tag], "-b", "128"] + extra) == 0
        err = capsys.readouterr().err
        assert ("--precision bf16x3" in err) == (tag == "bf16"), err
    a, b = open(outs["dsw"], "rb").read(), open(outs["ckpt"], "rb").read()
    assert a == b and a.count(b"\n") == n
    p32 = np.array([[float(x) for x in l.split("\t")[6:8]] for l in open(outs["dsw"])])
    p16 = np.array([[float(x) for x in l.split("\t")[6:8]] for l in open(outs["bf16"])])
    assert np.abs(p32 - p16).max() <= 5e-3 and np.abs(p16.sum(axis=1) - 1.0).max() <= 1e-6


def test_row_pipeline_fills_batches_across_items_with_identical_bits(small_weights):
    """call_mods' row pipeline (ds_submit / ds_wait, batches filled across queue items, several in flight): the rows it
    emits carry the bits of a plain blocking run over the same sites, in feed order, tags intact."""
    from deepsignal_amd import call_modifications as cm, fastio
    from deepsignal_amd.engine import Engine
    n = 1000
    feats = synth.synthetic_features(n, seed=3)
    eng = Engine(max_batch=128, slots=3)
    eng.load_weights(small_weights)
    ra, rp = eng.run(feats["kmer"], feats["means"], feats["stds"], feats["sanums"], feats["signals"])
    cuts = [0, 37, 37 + 128, 400, 401, 1000]          # ragged items: batches straddle item borders
    items = []
    for tag, (s0, e0) in enumerate(zip(cuts[:-1], cuts[1:])):
        m = e0 - s0
        info = np.frombuffer(("r%04d" * m % tuple(range(s0, e0))).encode(), np.uint8).copy()
        items.append(fastio.FeatureItem(info, np.arange(m + 1, dtype=np.int64) * 5, feats["kmer"][s0:e0], feats["means"][s0:e0],
                                        feats["stds"][s0:e0], feats["sanums"][s0:e0], feats["signals"][s0:e0],
                                        np.zeros(m, np.int32)))
    got = []
    pipe = cm._RowPipeline(eng, 128, lambda tag, data: got.append((tag, data)))
    for tag, it in enumerate(items):
        pipe.feed(it, tag)
    pipe.flush()
    assert not pipe.live_tags()
    pipe.close()
    eng.close()
    assert pipe.nsites == n and [t for t, _ in got] == sorted(t for t, _ in got)
    rows = b"".join(d for _, d in got).decode().splitlines()
    assert len(rows) == n
    info_all = np.frombuffer(("r%04d" * n % tuple(range(n))).encode(), np.uint8)
    expect = fastio.format_rows(info_all, np.arange(n + 1, dtype=np.int64) * 5, ra, rp, feats["kmer"]).decode().splitlines()
    assert rows == expect


def test_engine_after_a_failed_create_in_the_same_thread(small_weights, tmp_path, capsys):
    """An engine too large for the device fails in ds_create (hipMalloc) -- and the NEXT engine the same thread creates must
    work: every launcher of the library ends with hipGetLastError(), which would otherwise hand the stale out-of-memory error
    to that engine's first launch. Then the product's own use of it: make_engine's fall-back from an engine batch
    the GPU has no room for to the user's --batch_size."""
    from deepsignal_amd import call_modifications as cm
    from deepsignal_amd.engine import Engine
    from oracle import oracle
    feats = synth.synthetic_features(40, seed=77)
    with pytest.raises(RuntimeError, match="hipMalloc"):
        Engine(device=0, max_batch=1 << 21)
    eng = Engine(device=0, max_batch=64)
    eng.load_weights(small_weights)                 # finalize_weights launches embed_table_kernel: the first `return hipGetLastError()`
    act, pred = eng.run(feats["kmer"], feats["means"], feats["stds"], feats["sanums"], feats["signals"])
    eng.close()
    o_act, _ = oracle.forward(small_weights, feats, "f32")
    assert np.abs(act - o_act).max() <= 1e-5
    wfile = str(tmp_path / "model.dsw")
    W.save_weights(wfile, small_weights)
    eng = cm.make_engine(wfile, 17, 360, 2, 64, engine_batch=1 << 21)
    assert "falling back to --batch_size 64" in capsys.readouterr().err
    act2, _ = eng.run(feats["kmer"], feats["means"], feats["stds"], feats["sanums"], feats["signals"])
    eng.close()
    assert np.array_equal(act2, act)
