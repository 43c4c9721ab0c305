"""GPU: the BASELINE.json configurations that had no `-m gpu` leg in round 1, and the GPU legs of scope rows f2 / f3 /
f4:

  * configs[3]  per-read shard of >= 1 M synthetic sites on one GPU + the result gather (tools/config4.py's path),
                sampled sites against the oracle, size-independent properties over the whole shard;
  * configs[4]  fast5 -> extract_features (host) -> HIP engine -> rows, on the committed raw arrays of the
                reference-run extraction golden (no h5py needed): feature columns byte for byte against the
                reference extractor's rows, probabilities against the oracle-driven harness;
  * f3          a checkpoint assembled BY HAND in this file (LevelDB table + BundleEntryProto bytes, not through
                deepsignal_amd.tf_checkpoint's writer) -> `deepsignal call_mods -m <prefix>` on the GPU;
  * f4          HIP call_mods output -> call_modification_frequency, against the same from oracle output;
  * the committed tests/golden/forward_golden.npz replayed through the HIP path.
"""
import json
import os
import random
import struct
import sys

import numpy as np
import pytest

from deepsignal_amd import synth, weights as W

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
KEYS = ("kmer", "means", "stds", "sanums", "signals")
ACT_ATOL = 1e-5          # fp32 HIP path vs fp32 oracle (north-star gate: 1e-4 on normalised probabilities)


def _norm(act):
    return act / act.sum(axis=1, keepdims=True)


class OracleEngine:
    class_num = 2

    def __init__(self, weights):
        self.w = weights

    def run(self, kmer, means, stds, sanums, signals):
        from oracle import oracle
        feats = {"kmer": np.asarray(kmer, np.int32), "means": np.asarray(means, np.float32),
                 "stds": np.asarray(stds, np.float32), "sanums": np.asarray(sanums, np.float32),
                 "signals": np.asarray(signals, np.float32)}
        return oracle.forward(self.w, feats, "f32")


# ------------------------------------------------------------------------------------------------------------------
# tests/golden/forward_golden.npz through the HIP path
# ------------------------------------------------------------------------------------------------------------------
def test_forward_golden_vectors_replayed_on_the_gpu():
    """The committed golden vectors (float64 oracle, cross-checked by the independent PyTorch statement when they were
    generated) are the fixed point the HIP path is tied to: inputs from the file, weights from its seed, every
    recorded tensor compared."""
    from deepsignal_amd.engine import Engine
    g = np.load(os.path.join(GOLDEN, "forward_golden.npz"))
    w = W.random_weights(seed=int(g["weight_seed"]), lstm_bias_std=float(g["lstm_bias_std"]))
    n = g["in_kmer"].shape[0]
    eng = Engine(max_batch=16, debug=True)
    eng.load_weights(w)
    act, pred = eng.run(*(g["in_" + k] for k in KEYS))
    assert np.abs(act - g["act"]).max() <= ACT_ATOL
    decided = np.abs(g["act"][:, 1] - g["act"][:, 0]) > 1e-3
    assert (pred[decided] == g["pred"][decided]).all()

    def close(got, ref, name):
        tol = 2e-5 * max(1.0, float(np.abs(ref).max()))
        assert np.abs(got - ref).max() <= tol, name

    close(eng.intermediate("logits", (n, 2)), g["logits"], "logits")
    close(eng.intermediate("lstm_fw_l2", (n, 17, 256))[:, -1, :], g["lstm_fw_l2_last"], "lstm_fw_l2")
    close(eng.intermediate("lstm_bw_l2", (n, 17, 256))[:, 0, :], g["lstm_bw_l2_first"], "lstm_bw_l2")
    close(eng.intermediate("stem_pool", (n, 90, 64))[0], g["stem_pool_site0"], "stem_pool")
    close(eng.intermediate("module1", (n, 90, 240))[0], g["module1_site0"], "module1")
    close(eng.intermediate("module4", (n, 45, 240))[1], g["module4_site1"], "module4")
    close(eng.intermediate("module11", (n, 23, 240)), g["module11"], "module11")
    close(eng.intermediate("signal_feat", (n, 5520))[:, :512], g["signal_feat_head"], "signal_feat")
    close(eng.intermediate("fc1", (n, 6032))[:, :512], g["fc1_head"], "fc1")
    eng.close()
    # the product default (not debug): joint model folded into one 6032 x 2 matrix -- same golden outputs
    eng = Engine(max_batch=16)
    eng.load_weights(w)
    act, pred = eng.run(*(g["in_" + k] for k in KEYS))
    assert np.abs(act - g["act"]).max() <= ACT_ATOL and (pred[decided] == g["pred"][decided]).all()
    close(eng.intermediate("logits", (n, 2)), g["logits"], "logits (folded)")
    close(eng.intermediate("module11", (n, 23, 240)), g["module11"], "module11 (folded)")
    eng.close()


# ------------------------------------------------------------------------------------------------------------------
# BASELINE configs[3]
# ------------------------------------------------------------------------------------------------------------------
def test_config4_one_million_site_shard_on_one_gpu(small_weights):
    """1,048,560 sites (52,428 reads of 20) through ds_forward_device in 512-site forwards, 8 forwards in flight, then
    the result gather (world 1: the re-ordering path). Sampled sites against the oracle; over the WHOLE shard the
    size-independent properties: every output finite and a probability, pred = argmax, and a site's bits do not
    depend on where in the 2,048-step stream it ran (the pool of distinct sites repeats)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import config4
    from oracle import oracle
    sites, B, pool = 1_048_576, 512, 8192
    rec, g_act, g_pred, my_reads = config4.run_shard(sites, batch=B, precision="fp32", weights=small_weights, pool_sites=pool)
    total = (sites // 20) * 20
    assert rec["sites"] == total and rec["n_gpus"] == 1 and my_reads.size == sites // 20
    to_np = lambda x: x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)      # world 1 returns host arrays
    act, pred = to_np(g_act), to_np(g_pred)
    assert act.shape == (total, 2) and pred.shape == (total,)
    assert np.isfinite(act).all() and (act > 0).all() and (act < 1).all()
    assert np.array_equal(pred, np.argmax(act, axis=1).astype(np.int32))
    # site j of the stream ran the pool's site j % pool: every repetition must carry the same bits
    reps = total // pool
    a3 = act[:reps * pool].reshape(reps, pool, 2)
    assert np.array_equal(a3, np.broadcast_to(a3[0], a3.shape))
    assert np.array_equal(act[reps * pool:], act[:total - reps * pool])
    # sampled sites (from anywhere in the stream) against the oracle
    feats = config4.pool_features(pool)
    rng = np.random.default_rng(4)
    sel = np.sort(rng.choice(total, 768, replace=False))
    sub = {k: v[sel % pool] for k, v in feats.items()}
    o_act, o_pred = oracle.forward(small_weights, sub, "f32")
    assert np.abs(act[sel] - o_act).max() <= ACT_ATOL
    assert np.abs(_norm(act[sel]) - _norm(o_act)).max() <= 1e-4
    decided = np.abs(o_act[:, 1] - o_act[:, 0]) > 1e-3
    assert (pred[sel][decided] == o_pred[decided]).all()
    assert rec["sites_per_s"] > 0


# ------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4] / scope row f2
# ------------------------------------------------------------------------------------------------------------------
def _fake_read_fast5(g):
    def fake_read(path, corrected_group, basecall_subgroup):
        r = g["reads"][os.path.basename(path)[:-6]]
        return (np.asarray(r["signal"], np.int16), r["starts"], r["lengths"], r["bases"], r["range"] / r["digitisation"],
                r["offset"], (r["read_id"], r["strand"], r["alignstrand"], r["chrom"], r["chrom_start"]))
    return fake_read


def test_config5_fast5_arrays_to_rows_on_the_gpu(small_weights, tmp_path, monkeypatch):
    """call_mods on a fast5 directory: files -> host feature extraction (the committed raw arrays stand in for the HDF5
    access; the arithmetic is deepsignal_amd.extract_features) -> HIP engine -> result rows.
    (1) the features that reach the engine print to exactly the rows the REFERENCE extractor wrote for these reads;
    (2) the result rows equal the oracle-driven harness' rows: text columns byte for byte, probabilities to 1e-4."""
    from deepsignal_amd import call_modifications as cm, extract_features as ef
    g = json.load(open(os.path.join(GOLDEN, "extract_golden.json")))
    case = g["cases"][1]                       # zscore, CG, no reference genome (pos_in_strand = -1), k = 17, 360 samples
    d = tmp_path / "f5"
    d.mkdir()
    for name in g["read_order"]:
        (d / (name + ".fast5")).write_bytes(b"")
    monkeypatch.setattr(ef, "_read_fast5", _fake_read_fast5(g))
    f5_args = (True, "RawGenomeCorrected_000", "BaseCalled_template", None, True, case["normalize_method"], case["motifs"],
               0, 1, 2, None)
    # (1) what the extractor hands to the engine
    random.seed(case["seed"])
    fast5s = [str(d / (name + ".fast5")) for name in g["read_order"]]     # the order the reference run visited them in
    assert sorted(ef.get_fast5s(str(d), True)) == sorted(fast5s)
    feats_list, err = ef._extract_features(fast5s, "RawGenomeCorrected_000", "BaseCalled_template", case["normalize_method"],
                                           ef.get_motif_seqs(case["motifs"]), 0, None, 17, 360, 1, None)
    assert err == 0
    assert [ef._features_to_str(f) for f in feats_list] == case["features_str"]
    # (2) the same directory through call_mods: HIP engine (weights from a DSAMDW01 file) vs the oracle engine
    wfile = str(tmp_path / "model.dsw")
    W.save_weights(wfile, small_weights)
    out_gpu, out_cpu = str(tmp_path / "gpu.tsv"), str(tmp_path / "cpu.tsv")
    random.seed(case["seed"])
    n_gpu = cm.call_mods(str(d), wfile, out_gpu, 17, 360, 16, 0.001, 2, 1, True, True, True, True, f5_args)
    random.seed(case["seed"])
    n_cpu = cm.call_mods(str(d), wfile, out_cpu, 17, 360, 16, 0.001, 2, 1, True, True, True, True, f5_args,
                         engine=OracleEngine(small_weights))
    assert n_gpu == n_cpu == len(case["features_str"])
    rg = [l.rstrip("\n").split("\t") for l in open(out_gpu)]
    rc = [l.rstrip("\n").split("\t") for l in open(out_cpu)]
    ref7 = ["\t".join(r.split("\t")[:7]) for r in case["features_str"]]
    # sampleinfo + k-mer = the reference rows' columns (directory listing order is not the reference run's order)
    assert sorted("\t".join(r[:6] + [r[9]]) for r in rg) == sorted(ref7)
    for a, b in zip(rg, rc):
        assert len(a) == 10 and a[:6] == b[:6] and a[9] == b[9]
        assert abs(float(a[6]) - float(b[6])) <= 1e-4 and abs(float(a[7]) - float(b[7])) <= 1e-4
        if abs(float(b[6]) - float(b[7])) > 1e-3:
            assert a[8] == b[8]


@pytest.mark.parametrize("style", ["plain", "ont"])
def test_config5_real_fast5_files_opened_on_the_gpu_box(small_weights, tmp_path, style):
    """BASELINE configs[4] with the HDF5 access itself in the loop: `deepsignal call_mods -i <directory of .fast5 files>`
    on the committed h5py-written tombo-style files (tests/golden/fast5: contiguous + fixed strings, and the ONT storage
    form -- chunked + gzip + shuffle signal, variable-length strings). The box has no h5py: the files are opened by
    deepsignal_amd.minihdf5. Rows against the oracle-driven harness on the same files (text byte for byte,
    probabilities to 1e-4), and the features against the rows the REFERENCE extractor wrote for these reads."""
    import shutil
    from deepsignal_amd import call_modifications as cm, deepsignal as cli, extract_features as ef
    g = json.load(open(os.path.join(GOLDEN, "extract_golden.json")))
    case = g["cases"][1]                       # zscore, CG, no reference genome, k = 17, 360 samples
    d = tmp_path / "f5"
    shutil.copytree(os.path.join(GOLDEN, "fast5", style), str(d))
    wfile = str(tmp_path / "model.dsw")
    W.save_weights(wfile, small_weights)
    out_gpu, out_cpu = str(tmp_path / "gpu.tsv"), str(tmp_path / "cpu.tsv")
    random.seed(case["seed"])
    rc_ = cli.main(["call_mods", "-i", str(d), "-m", wfile, "-o", out_gpu, "--normalize_method", "zscore", "--f5_batch_num", "2",
                    "--batch_size", "16", "--is_gpu", "yes"])
    assert rc_ in (0, None)
    f5_args = (True, "RawGenomeCorrected_000", "BaseCalled_template", None, True, "zscore", "CG", 0, 1, 2, None)
    random.seed(case["seed"])
    n_cpu = cm.call_mods(str(d), wfile, out_cpu, 17, 360, 16, 0.001, 2, 1, True, True, True, True, f5_args,
                         engine=OracleEngine(small_weights))
    rg = [l.rstrip("\n").split("\t") for l in open(out_gpu)]
    rcpu = [l.rstrip("\n").split("\t") for l in open(out_cpu)]
    assert len(rg) == n_cpu == len(case["features_str"])
    ref7 = ["\t".join(r.split("\t")[:7]) for r in case["features_str"]]
    assert sorted("\t".join(r[:6] + [r[9]]) for r in rg) == sorted(ref7)
    for a, b in zip(rg, rcpu):
        assert len(a) == 10 and a[:6] == b[:6] and a[9] == b[9]
        assert abs(float(a[6]) - float(b[6])) <= 1e-4 and abs(float(a[7]) - float(b[7])) <= 1e-4
        if abs(float(b[6]) - float(b[7])) > 1e-3:
            assert a[8] == b[8]
    # which reader opened the files: the package's own one wherever h5py is absent (the MI355X image)
    try:
        import h5py  # noqa: F401
    except ImportError:
        assert ef._hdf5_module().__name__ == "deepsignal_amd.minihdf5"


# ------------------------------------------------------------------------------------------------------------------
# scope row f4
# ------------------------------------------------------------------------------------------------------------------
def _write_feature_tsv(path, feats, sampleinfo):
    from deepsignal_amd.utils.process_utils import code2base_dna
    with open(path, "w") as f:
        for i, info in enumerate(sampleinfo):
            f.write("\t".join([info, "".join(code2base_dna[int(c)] for c in feats["kmer"][i]),
                               ",".join("%s" % np.float32(x) for x in feats["means"][i]),
                               ",".join("%s" % np.float32(x) for x in feats["stds"][i]),
                               ",".join(str(int(x)) for x in feats["sanums"][i]),
                               ",".join("%s" % np.float32(x) for x in feats["signals"][i]), "1"]) + "\n")


def test_call_mods_output_feeds_modification_frequency(small_weights, tmp_path):
    """The step after the path: 600 calls over 40 genome positions (15 reads each) -> per-site frequency table. Built
    from the HIP engine's result file and from the oracle engine's: same sites, coverage and met / unmet counts
    (where no call sits on the 1e-3 decision margin), probability sums within 15 x 1e-4."""
    from deepsignal_amd import call_modification_frequency as cmf, call_modifications as cm
    from deepsignal_amd.engine import Engine
    n, npos = 600, 40
    feats = synth.synthetic_features(n, seed=77)
    info = ["chr%d\t%d\t%s\t%d\tread_%03d\tt" % (1 + (i % npos) % 3, 1000 + 7 * (i % npos), "+-"[(i % npos) % 2], 5000 - (i % npos),
                                                  i // npos) for i in range(n)]
    tsv = str(tmp_path / "features.tsv")
    _write_feature_tsv(tsv, feats, info)
    eng = Engine(max_batch=128)
    eng.load_weights(small_weights)
    out_gpu, out_cpu = str(tmp_path / "gpu.calls.tsv"), str(tmp_path / "cpu.calls.tsv")
    cm.call_mods(tsv, "unused", out_gpu, 17, 360, 128, 0.001, 2, 1, True, True, True, True, None, engine=eng)
    cm.call_mods(tsv, "unused", out_cpu, 17, 360, 128, 0.001, 2, 1, True, True, True, True, None,
                 engine=OracleEngine(small_weights))
    eng.close()
    fg, fc = str(tmp_path / "gpu.freq.tsv"), str(tmp_path / "cpu.freq.tsv")
    assert cmf.main(["-i", out_gpu, "-o", fg, "--sort"]) == 0 and cmf.main(["-i", out_cpu, "-o", fc, "--sort"]) == 0
    tg = [l.split("\t") for l in open(fg).read().splitlines()]
    tc = [l.split("\t") for l in open(fc).read().splitlines()]
    assert len(tg) == len(tc) == npos
    margin = {}
    for row in open(out_cpu):
        w_ = row.split("\t")
        k = (w_[0], w_[1])
        margin[k] = min(margin.get(k, 1.0), abs(float(w_[6]) - float(w_[7])))
    for a, b in zip(tg, tc):
        assert a[:4] == b[:4] and a[8] == b[8] == "15" and a[10] == b[10]      # site, strand, coverage, k-mer
        assert abs(float(a[4]) - float(b[4])) <= 15e-4 + 1e-3 and abs(float(a[5]) - float(b[5])) <= 15e-4 + 1e-3   # %.3f text
        if margin[(a[0], a[1])] > 1e-3:
            assert a[6:8] == b[6:8] and a[9] == b[9]                           # met, unmet, frequency
    # bedMethyl form of the same table
    assert cmf.main(["-i", out_gpu, "-o", fg + ".bed", "--sort", "--bed"]) == 0
    assert len(open(fg + ".bed").read().splitlines()) == npos


# ------------------------------------------------------------------------------------------------------------------
# scope row f3: a checkpoint assembled by hand (independent of deepsignal_amd.tf_checkpoint's writer)
# ------------------------------------------------------------------------------------------------------------------
from hand_checkpoint import hand_checkpoint as _hand_checkpoint  # noqa: E402  (tests/hand_checkpoint.py)


def test_cli_on_a_hand_assembled_tf_checkpoint(small_weights, tmp_path):
    """`deepsignal call_mods -m <checkpoint prefix>` on the GPU with a full-size (k = 17, 360 samples) checkpoint whose
    bytes this test assembled itself: the importer must deliver exactly the tensors that were put in (rows identical
    to an engine fed the weight dict directly), and the rows must agree with the oracle."""
    from deepsignal_amd import call_modifications as cm
    from deepsignal_amd.deepsignal import main
    from deepsignal_amd.engine import Engine
    prefix = str(tmp_path / "bn_17.sn_360.epoch_7.ckpt")
    tensors = dict(small_weights)
    some = [k for k in small_weights if k.endswith("kernel")][:6]
    for k in some:                                   # optimizer slots a real checkpoint carries; must be ignored
        tensors[k + "/Adam"] = np.zeros_like(small_weights[k])
        tensors[k + "/Adam_1"] = np.ones_like(small_weights[k])
    tensors["beta1_power"] = np.array([0.9], np.float32).reshape(())
    _hand_checkpoint(prefix, tensors)
    n = 200
    feats = synth.synthetic_features(n, seed=91)
    info = ["chr1\t%d\t+\t%d\tread_%02d\tt" % (100 + i, i, i // 20) for i in range(n)]
    tsv = str(tmp_path / "features.tsv")
    _write_feature_tsv(tsv, feats, info)
    out_ckpt, out_dict, out_cpu = str(tmp_path / "ckpt.tsv"), str(tmp_path / "dict.tsv"), str(tmp_path / "cpu.tsv")
    assert main(["call_mods", "-i", tsv, "-m", prefix, "-o", out_ckpt, "-b", "64", "--is_gpu", "yes"]) == 0
    eng = Engine(max_batch=64)
    eng.load_weights(small_weights)
    cm.call_mods(tsv, "unused", out_dict, 17, 360, 64, 0.001, 2, 1, True, True, True, True, None, engine=eng)
    eng.close()
    assert open(out_ckpt, "rb").read() == open(out_dict, "rb").read()
    cm.call_mods(tsv, "unused", out_cpu, 17, 360, 64, 0.001, 2, 1, True, True, True, True, None,
                 engine=OracleEngine(small_weights))
    for a, b in zip(open(out_ckpt), open(out_cpu)):
        a, b = a.split("\t"), b.split("\t")
        assert a[:6] == b[:6] and a[9] == b[9]
        assert abs(float(a[6]) - float(b[6])) <= 1e-4 and abs(float(a[7]) - float(b[7])) <= 1e-4


# ------------------------------------------------------------------------------------------------------------------
# rows 8c / f3 against TensorFlow's own output (fixture of tests/golden/make_tf_golden.py; skipped loudly until it exists)
# ------------------------------------------------------------------------------------------------------------------
def test_tf_written_checkpoint_on_the_gpu():
    """The HIP engine fed the checkpoint TensorFlow wrote (through deepsignal_amd.tf_checkpoint; from the seed when only
    the npz is present) against TensorFlow's own activation_logits / prediction on the same feed: normalised
    probabilities within the north star's 1e-4, labels equal wherever the margin exceeds 1e-3 -- both joint-model forms, and
    both fp32-class precisions: DS_PRECISION_BF16X3 consumes the same inputs and the same fp32 weights (its terms are formed
    inside the engine), so the day the fixture exists it pins that mode to TensorFlow with the same bars."""
    import tf_golden_fixture as fx
    from deepsignal_amd import tf_checkpoint
    from deepsignal_amd.engine import Engine
    g = fx.golden()
    feats = fx.features_of(g)
    prefix = os.path.join(fx.TF_DIR, "model.ckpt")
    w = tf_checkpoint.checkpoint_to_weights(prefix) if tf_checkpoint.is_checkpoint(prefix) else fx.weights_of(g)
    tf_act, tf_pred = g["act"], g["pred"]
    decided = np.abs(tf_act[:, 1] - tf_act[:, 0]) > 1e-3
    for precision in ("fp32", "bf16x3"):
        for fold in (True, False):
            eng = Engine(max_batch=16, fold_fc=fold, precision=precision)
            eng.load_weights(w)
            act, pred = eng.run(*(feats[k] for k in KEYS))
            eng.close()
            assert np.abs(act - tf_act).max() <= 2e-5 and np.abs(_norm(act) - _norm(tf_act)).max() <= 1e-4, (precision, fold)
            assert (pred[decided] == tf_pred[decided]).all(), (precision, fold)
