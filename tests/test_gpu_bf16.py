"""GPU tests of the mixed-precision mode (BASELINE.json configs[2]: "bf16 conv+FC with fp32 BiLSTM accumulate,
batch=4096, tolerance vs fp32 reported"; include/deepsignal_hip.h DS_PRECISION_BF16).

Two bars:
  * exactness of the implementation: against oracle/torch_statement.forward_bf16, which rounds to bf16 at the same
    points (weights after BN folding, every stored conv activation, the FC operand). Only the fp32 accumulation
    order differs, which can flip an occasional bf16 rounding (and a flipped input moves downstream values), so
    the bound on the intermediates is: worst element within 4 bf16 ulps of the tensor's largest value, mean
    difference below 1/4 ulp (measured <= 0.09 ulp at module 11), and 3e-3 on the sigmoid outputs. A first flip is
    rare (~1e-5 per element) but cascades: sites without one reproduce the statement to ~1e-7, sites with one end up
    a different realisation of the same bf16 rounding noise (~1e-3 on the outputs, the size of the bf16-vs-fp32
    distance itself). The test prints the share of sites that match to 1e-6.
  * tolerance vs fp32 (the number configs[2] asks to report): max |p_bf16 - p_fp32| <= 5e-3 at batch 4096 and
    equal labels wherever the fp32 margin |p1 - p0| exceeds 2e-2. The measured value is printed (-s) and recorded
    in DESIGN.md.
"""
import numpy as np
import pytest

from oracle import torch_statement
from deepsignal_amd import synth

pytestmark = pytest.mark.gpu

EMU_TAP_TOL_ULPS = 4.0
EMU_TAP_MEAN_ULPS = 0.25


def _ulp(x):
    """bf16 spacing (8 significant bits) at magnitude x"""
    return 2.0 ** (np.floor(np.log2(max(float(x), 1e-30))) - 7)

EMU_ACT_ATOL = 3e-3
FP32_ACT_ATOL = 5e-3
FP32_LABEL_MARGIN = 2e-2


def _engine(weights, **kw):
    from deepsignal_amd.engine import Engine
    eng = Engine(**kw)
    eng.load_weights(weights)
    return eng


@pytest.mark.parametrize("precision", ["bf16", "bf16_all"])
@pytest.mark.parametrize("n", [1, 24, 130])
def test_bf16_layerwise_vs_emulated_statement(small_weights, n, precision):
    feats = synth.synthetic_features(n, seed=300 + n)
    eng = _engine(small_weights, max_batch=160, debug=True, precision=precision)
    act, pred = eng.run(feats["kmer"], feats["means"], feats["stds"], feats["sanums"], feats["signals"])
    e_act, e_pred, taps = torch_statement.forward_bf16(small_weights, feats, return_taps=True,
                                                       lstm_bf16=precision == "bf16_all")
    bad = {}
    for name, ref in taps.items():
        got = eng.intermediate(name, ref.shape)
        err = float(np.abs(got - ref).max())
        if (name.startswith("lstm_") and precision == "bf16") or name in ("fc1", "logits"):
            # fp32 tensors: the BiLSTM is untouched in "bf16"; fc1 / logits see the flips of their bf16 inputs
            tol = (2e-5 if name.startswith("lstm_") else 1e-2) * max(1.0, float(np.abs(ref).max()))
        else:
            u = _ulp(max(1.0, float(np.abs(ref).max())))
            tol = EMU_TAP_TOL_ULPS * u
            if not float(np.abs(got - ref).mean()) <= EMU_TAP_MEAN_ULPS * u:
                bad[name + ":mean"] = (float(np.abs(got - ref).mean()), EMU_TAP_MEAN_ULPS * u)
        if not err <= tol:
            bad[name] = (err, tol)
        if n == 24 and precision == "bf16":
            print("%-14s max|d| %.3e  mean|d| %.3e  max|ref| %.3f" % (name, err, float(np.abs(got - ref).mean()), float(np.abs(ref).max())))
    assert not bad, "bf16 intermediates out of tolerance: %s" % bad
    assert np.isfinite(act).all()
    assert np.abs(act - e_act).max() <= EMU_ACT_ATOL
    print("n=%d: %.0f%% of the sites reproduce the emulated statement to 1e-6, worst |d act| = %.2e"
          % (n, 100.0 * float((np.abs(act - e_act).max(axis=1) < 1e-6).mean()), float(np.abs(act - e_act).max())))
    eng.close()


@pytest.mark.parametrize("precision", ["bf16", "bf16_all"])
def test_bf16_config3_tolerance_vs_fp32(small_weights, precision):
    n = 4096
    feats = synth.synthetic_features(n, seed=4096)
    args = [feats[k] for k in ("kmer", "means", "stds", "sanums", "signals")]
    e32 = _engine(small_weights, max_batch=n, slots=1)
    a32, p32 = e32.run(*args)
    e32.close()
    e16 = _engine(small_weights, max_batch=n, slots=1, precision=precision)
    a16, p16 = e16.run(*args)
    # ragged / small batches go through the non-dense kernel variants and must agree with the big batch
    a_small, p_small = e16.run(*(a[100:177] for a in args))
    e16.close()
    assert np.isfinite(a16).all()
    diff = float(np.abs(a16 - a32).max())
    decided = np.abs(a32[:, 1] - a32[:, 0]) > FP32_LABEL_MARGIN
    agree = float((p16[decided] == p32[decided]).mean()) if decided.any() else 1.0
    pn = lambda a: a / a.sum(axis=1, keepdims=True)
    print("\n%s vs fp32 at batch %d: max|d act| = %.3e, mean|d act| = %.3e, max|d p_norm| = %.3e, labels equal on %.4f of %d "
          "decided sites, label flips overall %.5f"
          % (precision, n, diff, float(np.abs(a16 - a32).mean()), float(np.abs(pn(a16) - pn(a32)).max()), agree,
             int(decided.sum()), float((p16 != p32).mean())))
    assert diff <= FP32_ACT_ATOL
    assert agree == 1.0
    assert np.array_equal(a_small, a16[100:177]) and np.array_equal(p_small, p16[100:177])


# bf16 distance to fp32 where the read-out is balanced (tests/golden/make_stress_golden.py): (mean |d act| gate, share of
# label flips allowed, fp32 margin |p1 - p0| above which no label may flip). Measured values: DESIGN.md section 9.
TRAINED_REGIME_GATES = {"balanced": (0.04, 0.10, 0.30), "stress": (0.06, 0.10, 0.60)}


@pytest.mark.parametrize("precision", ["bf16", "bf16_all"])
@pytest.mark.parametrize("which", ["balanced", "stress"])
def test_bf16_config3_tolerance_vs_fp32_in_the_trained_regime(request, which, precision):
    """configs[2]'s "tolerance vs fp32 reported", second column: on `small_weights` every logit is within +-0.7, every site
    has the same label and '0 flips' says nothing. Here the head is centred (both labels, anti-correlated columns; `stress`
    also saturates the LSTM and has hot BN channels): the read-out then amplifies the site-to-site variation of features
    whose common mode is ~13x larger, and bf16 storage noise (2^-9 per stored activation, through 3 + 11 x 6 layers) shows.
    The numbers are a property of bf16 storage, not of this implementation: the CPU emulation that rounds at the same points
    (oracle/torch_statement.forward_bf16) is held to the same statistics on a 256-site subset."""
    import json
    import os
    n = 4096
    w = request.getfixturevalue(which + "_weights")
    feats = synth.synthetic_features(n, seed=4096)
    args = [feats[k] for k in ("kmer", "means", "stds", "sanums", "signals")]
    e32 = _engine(w, max_batch=n, slots=1)
    a32, p32 = e32.run(*args)
    e32.close()
    e16 = _engine(w, max_batch=n, slots=1, precision=precision)
    a16, p16 = e16.run(*args)
    e16.close()
    assert np.isfinite(a16).all()
    pn = lambda a: a / a.sum(axis=1, keepdims=True)
    d = np.abs(a16 - a32).max(axis=1)
    dpn = np.abs(pn(a16) - pn(a32)).max(axis=1)
    margin = np.abs(a32[:, 1] - a32[:, 0])
    flips = p16 != p32
    sub = {k: v[:256] for k, v in feats.items()}
    e_act, e_pred = torch_statement.forward_bf16(w, sub, lstm_bf16=precision == "bf16_all")[:2]
    d_emu = np.abs(e_act - a32[:256]).max(axis=1)
    rec = {"n": n, "max_abs_d_act": float(d.max()), "mean_abs_d_act": float(d.mean()), "p99_abs_d_act": float(np.quantile(d, 0.99)),
           "max_abs_d_pnorm": float(dpn.max()), "mean_abs_d_pnorm": float(dpn.mean()),
           "label_flip_rate": float(flips.mean()), "largest_fp32_margin_of_a_flipped_site": float(margin[flips].max()) if flips.any() else 0.0,
           "label1_share_fp32": float(p32.mean()),
           "cpu_emulation_256": {"max_abs_d_act": float(d_emu.max()), "mean_abs_d_act": float(d_emu.mean()),
                                 "label_flip_rate": float((e_pred != p32[:256]).mean())},
           "engine_vs_emulation_256_mean_abs": float(np.abs(a16[:256] - e_act).max(axis=1).mean())}
    print("\n%s / %s vs fp32 at batch %d: %s" % (which, precision, n, json.dumps(rec)))
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "stress_bf16_tolerance.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        old = json.load(open(path)) if os.path.exists(path) else {}
        old["%s_%s" % (which, precision)] = rec
        json.dump(old, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
    assert 0.2 <= float(p32.mean()) <= 0.8
    mean_gate, flip_gate, margin_gate = TRAINED_REGIME_GATES[which]
    assert rec["mean_abs_d_act"] <= mean_gate
    assert rec["label_flip_rate"] <= flip_gate
    assert rec["largest_fp32_margin_of_a_flipped_site"] <= margin_gate
    # the engine is as far from fp32 as the emulation of its rounding points is -- not further
    assert float(d[:256].mean()) <= 1.5 * float(d_emu.mean()) + 1e-3


def test_bf16_against_oracle_small(small_weights):
    """Same gate as the north star states for the path (outputs within 1e-4 of the reference) does NOT hold for
    bf16 -- this test documents the actual distance to the fp32 CPU oracle on a small batch."""
    from oracle import oracle
    feats = synth.synthetic_features(64, seed=77)
    eng = _engine(small_weights, max_batch=64, precision="bf16")
    act, pred = eng.run(feats["kmer"], feats["means"], feats["stds"], feats["sanums"], feats["signals"])
    o_act, o_pred = oracle.forward(small_weights, feats, "f32")
    assert np.abs(act - o_act).max() <= FP32_ACT_ATOL
    eng.close()


@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16_all"])
def test_module_chain_gives_the_bits_of_one_launch_per_module(small_weights, precision):
    """Every precision takes a tile of whole sites through all modules of a width class inside ONE launch (modules 1-3,
    4-8, 9-11); DS_TUNE_NO_CHAIN launches every module on its own. Same arithmetic, same bits -- also on a ragged batch
    whose last tile is partial, and with the per-module tap buffers of debug mode."""
    feats = synth.synthetic_features(1333, seed=812)
    args = [feats[k] for k in ("kmer", "means", "stds", "sanums", "signals")]
    chained = _engine(small_weights, max_batch=1333, slots=1, precision=precision)
    a1, p1 = chained.run(*args)
    a1s, p1s = chained.run(*(a[:77] for a in args))
    chained.close()
    single = _engine(small_weights, max_batch=1333, slots=1, precision=precision, chain_modules=False)
    a2, p2 = single.run(*args)
    a2s, p2s = single.run(*(a[:77] for a in args))
    single.close()
    assert np.array_equal(a1, a2) and np.array_equal(p1, p2)
    assert np.array_equal(a1s, a2s) and np.array_equal(p1s, p2s) and np.array_equal(a1s, a1[:77])
    dbg = _engine(small_weights, max_batch=160, debug=True, precision=precision)
    a3, _ = dbg.run(*(a[:130] for a in args))
    m11 = dbg.intermediate("module11", (130, 23, 240))
    dbg.close()
    # debug mode runs the three-step joint model (bf16-operand dense layer): the same sites, a rounding apart
    assert np.isfinite(m11).all() and np.abs(a3 - a1[:130]).max() <= 2e-3


def test_bf16_lstm_tile_shapes_give_the_same_bits(small_weights):
    """lstm_cell_bf16_kernel runs 64 x 64, 64 x 128 or 128 x 128 workgroup tiles by forward size (<= 768, <= 2047, more):
    every shape accumulates a unit's K in the same order, so a site's bits do not depend on the batch it travels in --
    also with an odd number of 32-site m-tiles (2100 sites = 66 m-tiles: the last 128-site block is half empty)."""
    feats = synth.synthetic_features(2100, seed=905)
    args = [feats[k] for k in ("kmer", "means", "stds", "sanums", "signals")]
    eng = _engine(small_weights, max_batch=2100, slots=1, precision="bf16_all")
    a_big, p_big = eng.run(*args)                          # 128 x 128 tiles
    a_mid, p_mid = eng.run(*(a[:1000] for a in args))      # 64 x 128
    a_small, p_small = eng.run(*(a[:700] for a in args))   # 64 x 64
    a_one, p_one = eng.run(*(a[2099:2100] for a in args))
    eng.close()
    assert np.isfinite(a_big).all()
    assert np.array_equal(a_mid, a_big[:1000]) and np.array_equal(p_mid, p_big[:1000])
    assert np.array_equal(a_small, a_big[:700]) and np.array_equal(p_small, p_big[:700])
    assert np.array_equal(a_one, a_big[2099:2100]) and np.array_equal(p_one, p_big[2099:2100])


@pytest.mark.parametrize("precision", ["bf16", "bf16_all"])
@pytest.mark.parametrize("variant", [dict(), dict(is_cnn=False), dict(is_rnn=False), dict(class_num=3)])
def test_bf16_folded_head_against_the_three_step_bf16_path(variant, precision):
    """bf16 modes, default engine: the head multiplies [bf16 h_fw | bf16 h_bw | module 11's bf16 rows] with the fp32 matrix
    (avgpool^T)(W1 W2) -- no pooling kernel, no rounding of the pooled features. The three-step form of the same precision
    (DS_TUNE_NO_FOLD_FC: bf16 pooling kernel, bf16 dense(J, J) weights, fp32 dense(J, C)) is the same function up to those
    bf16 roundings: logits within 2e-2 of their scale, sigmoids within 5e-3, equal labels wherever the margin exceeds 2e-2."""
    from deepsignal_amd import weights as W
    w = W.random_weights(seed=21, lstm_bias_std=0.1, **variant)
    feats = synth.synthetic_features(200, seed=78)
    keys = ("kmer", "means", "stds", "sanums", "signals")
    folded = _engine(w, max_batch=256, precision=precision, **variant)
    steps = _engine(w, max_batch=256, precision=precision, fold_fc=False, **variant)
    act_f, pred_f = folded.run(*[feats[k] for k in keys])
    act_s, pred_s = steps.run(*[feats[k] for k in keys])
    C = variant.get("class_num", 2)
    lf, ls = folded.intermediate("logits", (200, C)), steps.intermediate("logits", (200, C))
    scale = max(1.0, float(np.abs(ls).max()))
    print("bf16 folded vs three-step (%s, %s): max |d logit| %.3e (scale %.2f), max |d act| %.3e" %
          (precision, variant, float(np.abs(lf - ls).max()), scale, float(np.abs(act_f - act_s).max())))
    assert np.abs(lf - ls).max() <= 2e-2 * scale
    assert np.abs(act_f - act_s).max() <= FP32_ACT_ATOL
    srt = np.sort(act_s, axis=1)
    decided = srt[:, -1] - srt[:, -2] > FP32_LABEL_MARGIN
    assert (pred_f[decided] == pred_s[decided]).all()
    folded.close(); steps.close()


@pytest.mark.parametrize("geom,batch", [(dict(kmer_len=17, signal_len=128), 1024), (dict(kmer_len=9, signal_len=62), 1536),
                                        (dict(kmer_len=17, signal_len=360), 700)])
def test_bf16_all_other_geometries_and_ragged_batches(geom, batch):
    """--cent_signals_len / --kmer_len other than the defaults change every tile shape of the bf16 kernels (module widths
    32 / 16 / 8 and 16 / 8 / 4 instead of 90 / 45 / 23: up to twelve sites per 96-row tile, chains of the same lengths) and
    the BiLSTM's step count; 700 sites leave ragged last tiles everywhere. Outputs within the bf16 tolerance of the fp32
    engine, and a site's bits must not depend on the batch it travels in."""
    from deepsignal_amd import weights as W
    w = W.random_weights(seed=36, lstm_bias_std=0.1, **geom)
    feats = synth.synthetic_features(batch, seed=10, **geom)
    keys = ("kmer", "means", "stds", "sanums", "signals")
    f32 = _engine(w, max_batch=batch, **geom)
    r_act, r_pred = f32.run(*(feats[k] for k in keys))
    f32.close()
    eng = _engine(w, max_batch=batch, precision="bf16_all", **geom)
    act, pred = eng.run(*(feats[k] for k in keys))
    assert np.isfinite(act).all()
    print("bf16_all %s batch %d: max |d act| vs fp32 %.3e" % (geom, batch, float(np.abs(act - r_act).max())))
    assert np.abs(act - r_act).max() <= FP32_ACT_ATOL
    srt = np.sort(r_act, axis=1)
    decided = srt[:, -1] - srt[:, -2] > FP32_LABEL_MARGIN
    assert (pred[decided] == r_pred[decided]).all()
    sel = np.random.default_rng(2).choice(batch, 83, replace=False)
    a2, p2 = eng.run(*(feats[k][sel] for k in keys))
    assert np.array_equal(a2, act[sel]) and np.array_equal(p2, pred[sel])
    eng.close()



@pytest.mark.parametrize("precision", ["bf16", "bf16_all"])
def test_bf16_layerwise_vs_emulated_statement_on_the_stress_set(stress_weights, precision):
    """The exactness bar of the first test of this file -- every stored tensor within a few bf16 ulps of the CPU statement that
    rounds at the engine's rounding points -- on the trained-regime weights: larger activations (hot BN channels), saturating
    LSTM gates. The sigmoid outputs are NOT held to 3e-3 here: one flipped bf16 rounding upstream moves a centred read-out by
    ~1e-2 (that is the tolerance-vs-fp32 story of the test above); the taps are what shows the implementation is right."""
    n = 64
    feats = synth.synthetic_features(n, seed=321)
    eng = _engine(stress_weights, max_batch=160, debug=True, precision=precision)
    act, pred = eng.run(feats["kmer"], feats["means"], feats["stds"], feats["sanums"], feats["signals"])
    e_act, e_pred, taps = torch_statement.forward_bf16(stress_weights, feats, return_taps=True, lstm_bf16=precision == "bf16_all")
    bad, worst = {}, {}
    for name, ref in taps.items():
        if name in ("fc1", "logits", "joint"):
            continue
        got = eng.intermediate(name, ref.shape)
        err = float(np.abs(got - ref).max())
        if name.startswith("lstm_") and precision == "bf16":
            tol, mean_tol = 5e-5, None                                   # fp32 BiLSTM (saturating: fp32 noise ~1e-5)
        else:
            u = _ulp(max(1.0, float(np.abs(ref).max())))
            tol, mean_tol = EMU_TAP_TOL_ULPS * u, EMU_TAP_MEAN_ULPS * u
        worst[name] = (err, tol)
        if not err <= tol:
            bad[name] = (err, tol)
        if mean_tol is not None and not float(np.abs(got - ref).mean()) <= mean_tol:
            bad[name + ":mean"] = (float(np.abs(got - ref).mean()), mean_tol)
    print("\n%s stress taps (max err, tol):" % precision, {k: ("%.2e" % v[0], "%.2e" % v[1]) for k, v in worst.items() if k in ("stem_conv3", "module1", "module6", "module11", "lstm_fw_l2")})
    print("   outputs: max |d act| vs the emulation %.3e, share of sites within 1e-6: %.2f" % (float(np.abs(act - e_act).max()), float((np.abs(act - e_act).max(axis=1) < 1e-6).mean())))
    assert not bad, bad
    assert np.isfinite(act).all() and float(np.abs(act - e_act).max()) <= 0.1
    eng.close()
