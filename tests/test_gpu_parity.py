"""GPU parity: HIP engine (through the C ABI) vs the CPU oracle on the same seeded inputs.

Tolerances (fp32 everywhere; BASELINE.json north_star: outputs within 1e-4 of the reference):
  * intermediates: max |gpu - oracle_f32| <= 2e-5 * max(1, max|oracle|)   (different summation order only)
  * act (sigmoid outputs) and normalised probabilities: <= 1e-5 absolute (10x inside the 1e-4 gate)
  * labels equal wherever |p1 - p0| > 1e-3
"""
import numpy as np
import pytest

from deepsignal_amd import spec, synth

pytestmark = pytest.mark.gpu

INTERMEDIATE_RTOL = 2e-5
ACT_ATOL = 1e-5
# DS_PRECISION_BF16X3 (fp32 operands as three bf16 terms on the bf16 matrix pipe, six products per MAC) is held to the SAME
# bars as native fp32: the tolerances above are not widened for it.
FP32_CLASS = ["fp32", "bf16x3"]


def _engine(weights, **kw):
    from deepsignal_amd.engine import Engine
    eng = Engine(**kw)
    eng.load_weights(weights)
    return eng


def _norm(act):
    return act / act.sum(axis=1, keepdims=True)


def _check_outputs(act, pred, o_act, o_pred):
    assert np.isfinite(act).all()
    assert np.abs(act - o_act).max() <= ACT_ATOL
    assert np.abs(_norm(act) - _norm(o_act)).max() <= ACT_ATOL
    decided = np.abs(o_act[:, 1] - o_act[:, 0]) > 1e-3
    assert (pred[decided] == o_pred[decided]).all()


@pytest.mark.parametrize("precision", FP32_CLASS)
@pytest.mark.parametrize("n", [1, 24, 130])
def test_layerwise_parity_vs_oracle(small_weights, n, precision):
    from oracle import oracle
    feats = synth.synthetic_features(n, seed=100 + n)
    eng = _engine(small_weights, max_batch=160, debug=True, precision=precision)
    act, pred = eng.run(feats["kmer"], feats["means"], feats["stds"], feats["sanums"], feats["signals"])
    o_act, o_pred, taps = oracle.forward(small_weights, feats, "f32", taps=True)
    worst = {}
    for name, ref in taps.items():
        got = eng.intermediate(name, ref.shape)
        err = float(np.abs(got - ref).max())
        tol = INTERMEDIATE_RTOL * max(1.0, float(np.abs(ref).max()))
        worst[name] = (err, tol)
    bad = {k: v for k, v in worst.items() if not v[0] <= v[1]}
    assert not bad, "intermediates out of tolerance: %s" % bad
    _check_outputs(act, pred, o_act, o_pred)
    eng.close()


@pytest.mark.parametrize("precision", FP32_CLASS)
def test_batch_512_and_ragged_tail(small_weights, precision):
    """n > max_batch is looped inside ds_forward; last chunk is partial (call_modifications.py:157-166)."""
    from oracle import oracle
    n = 512 + 37
    feats = synth.synthetic_features(n, seed=5)
    eng = _engine(small_weights, max_batch=512, precision=precision)
    act, pred = eng.run(feats["kmer"], feats["means"], feats["stds"], feats["sanums"], feats["signals"])
    sel = np.r_[0:40, 500:549]
    sub = {k: v[sel] for k, v in feats.items()}
    o_act, o_pred = oracle.forward(small_weights, sub, "f32")
    _check_outputs(act[sel], pred[sel], o_act, o_pred)
    # batch-composition independence: the same site gives bit-identical results alone and in a batch
    a1, p1 = eng.run(*(feats[k][7:8] for k in ("kmer", "means", "stds", "sanums", "signals")))
    assert np.array_equal(a1[0], act[7]) and p1[0] == pred[7]
    eng.close()


def test_dense_and_masked_kernel_variants_agree_bitwise(small_weights):
    """Batches that are a multiple of 128 take the row-mask-free kernel instantiations; a site must get the same
    bits either way (explicit fma chains in the LSTM epilogue keep the two instantiations' rounding identical)."""
    feats = synth.synthetic_features(256, seed=12)
    keys = ("kmer", "means", "stds", "sanums", "signals")
    eng = _engine(small_weights, max_batch=256)
    act, pred = eng.run(*(feats[k] for k in keys))
    a2, p2 = eng.run(*(feats[k][100:177] for k in keys))
    assert np.array_equal(a2, act[100:177]) and np.array_equal(p2, pred[100:177])
    eng.close()


def test_graph_and_eager_agree_and_are_deterministic(small_weights):
    feats = synth.synthetic_features(96, seed=9)
    eng = _engine(small_weights, max_batch=128)
    args = [feats[k] for k in ("kmer", "means", "stds", "sanums", "signals")]
    a_g, p_g = eng.run(*args)
    a_g2, _ = eng.run(*args)
    eng.set_graph(False)
    a_e, p_e = eng.run(*args)
    eng.set_profiling(True)
    a_p, _ = eng.run(*args)
    assert np.array_equal(a_g, a_g2) and np.array_equal(a_g, a_e) and np.array_equal(a_g, a_p)
    assert np.array_equal(p_g, p_e)
    st = {s["name"]: s for s in eng.stage_times()}
    # the default engine folds the joint model (no fc1 stage); layer-0 input projection is a table lookup, so the LSTM
    # does fewer MACs than the reference graph
    assert "fc1" not in st and st["head"]["calls"] == 1 and st["head"]["total_ms"] > 0
    total = sum(s["flops_per_site"] for s in st.values())
    folded = spec.FLOPS_PER_SITE - 2.0 * 6032 * 6032
    assert 0.9 * folded < total <= folded * 1.001
    eng.close()


@pytest.mark.parametrize("precision", FP32_CLASS)
def test_edge_inputs(small_weights, precision):
    """All-N k-mers, zero-padded (short) signal windows, extreme event lengths, empty batch."""
    from oracle import oracle
    feats = synth.synthetic_features(16, seed=11)
    feats["kmer"][0, :] = 4
    feats["signals"][1, 17:] = 0.0
    feats["signals"][2, :] = 0.0
    feats["sanums"][3, :] = 200.0
    feats["means"][4, :] = 5.0
    feats["means"][5, :] = -5.0
    eng = _engine(small_weights, max_batch=64, precision=precision)
    args = [feats[k] for k in ("kmer", "means", "stds", "sanums", "signals")]
    act, pred = eng.run(*args)
    o_act, o_pred = oracle.forward(small_weights, feats, "f32")
    _check_outputs(act, pred, o_act, o_pred)
    e_act, e_pred = eng.run(*(a[:0] for a in args))
    assert e_act.shape == (0, 2) and e_pred.shape == (0,)
    eng.close()


def test_weight_file_roundtrip_matches_in_memory(small_weights, tmp_path):
    from deepsignal_amd import weights as W
    from deepsignal_amd.engine import Engine
    path = str(tmp_path / "w.dsw")
    W.save_weights(path, small_weights)
    feats = synth.synthetic_features(8, seed=13)
    args = [feats[k] for k in ("kmer", "means", "stds", "sanums", "signals")]
    e1 = _engine(small_weights, max_batch=32)
    e2 = Engine(max_batch=32)
    e2.load_weights_file(path)
    a1, _ = e1.run(*args)
    a2, _ = e2.run(*args)
    assert np.array_equal(a1, a2)
    e1.close(); e2.close()


def test_errors_are_loud(small_weights):
    from deepsignal_amd.engine import Engine
    eng = Engine(max_batch=8)
    feats = synth.synthetic_features(4, seed=1)
    with pytest.raises(RuntimeError):        # weights not loaded
        eng.run(*(feats[k] for k in ("kmer", "means", "stds", "sanums", "signals")))
    with pytest.raises(RuntimeError):        # model.py:28-29: at least one of is_cnn / is_rnn
        Engine(is_cnn=False, is_rnn=False)
    bad = dict(small_weights)
    bad.pop("dense/kernel")
    with pytest.raises(RuntimeError):
        eng.load_weights(bad)
    eng.close()


@pytest.mark.parametrize("variant", [dict(is_cnn=False, is_rnn=True, is_base=True),     # RNN-only (what `denoise` uses)
                                     dict(is_cnn=True, is_rnn=False, is_base=True),     # CNN-only
                                     dict(is_cnn=True, is_rnn=True, is_base=False),     # no k-mer embedding
                                     dict(is_cnn=False, is_rnn=True, is_base=False)])
@pytest.mark.parametrize("precision", FP32_CLASS)
def test_model_variants(variant, precision):
    """Model(is_cnn, is_rnn, is_base) switches of model.py:28-29,59-75,89-95."""
    from deepsignal_amd import weights as W
    from oracle import oracle
    w = W.random_weights(seed=21, lstm_bias_std=0.1, **variant)
    feats = synth.synthetic_features(40, seed=77)
    eng = _engine(w, max_batch=64, debug=True, precision=precision, **variant)
    act, pred = eng.run(feats["kmer"], feats["means"], feats["stds"], feats["sanums"], feats["signals"])
    o_act, o_pred, taps = oracle.forward(w, feats, "f32", taps=True, **variant)
    for name, ref in taps.items():
        got = eng.intermediate(name, ref.shape)
        err = float(np.abs(got - ref).max())
        assert err <= INTERMEDIATE_RTOL * max(1.0, float(np.abs(ref).max())), (name, err)
    _check_outputs(act, pred, o_act, o_pred)
    eng.close()


@pytest.mark.parametrize("precision", FP32_CLASS)
def test_pipelined_device_forwards_match_blocking(small_weights, precision):
    """ds_forward_device rotates over independent slots (several forwards in flight): every slot must give
    bit-identical results to the blocking host path, in any interleaving."""
    import torch
    feats = synth.synthetic_features(4 * 512, seed=31)
    eng = _engine(small_weights, max_batch=512, slots=5, precision=precision)
    dev = torch.device("cuda", 0)
    d = {k: torch.from_numpy(feats[k]).to(dev) for k in ("kmer", "means", "stds", "sanums", "signals")}
    nstep = 13
    out_act = torch.zeros((nstep, 512, 2), dtype=torch.float32, device=dev)
    out_pred = torch.zeros((nstep, 512), dtype=torch.int32, device=dev)
    for i in range(nstep):
        b = (i % 4) * 512
        eng.run_device(512, *(d[k][b:b + 512].data_ptr() for k in ("kmer", "means", "stds", "sanums", "signals")),
                       out_act[i].data_ptr(), out_pred[i].data_ptr())
    eng.sync()
    ref = [eng.run(*(feats[k][j * 512:(j + 1) * 512] for k in ("kmer", "means", "stds", "sanums", "signals"))) for j in range(4)]
    got_act, got_pred = out_act.cpu().numpy(), out_pred.cpu().numpy()
    for i in range(nstep):
        assert np.array_equal(got_act[i], ref[i % 4][0]) and np.array_equal(got_pred[i], ref[i % 4][1])
    # spot-check the batch against the oracle too
    from oracle import oracle
    sub = {k: v[512:512 + 64] for k, v in feats.items()}
    o_act, o_pred = oracle.forward(small_weights, sub, "f32")
    _check_outputs(got_act[1][:64], got_pred[1][:64], o_act, o_pred)
    eng.close()


@pytest.mark.parametrize("geom", [dict(kmer_len=9, signal_len=100), dict(kmer_len=21, signal_len=128),
                                  dict(kmer_len=5, signal_len=40)])
@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "bf16_all"])
def test_other_kmer_and_signal_lengths(geom, precision):
    """--kmer_len / --cent_signals_len are free parameters of the reference CLI (deepsignal.py:258-263): widths,
    SAME paddings, the joint width and every tile shape follow them."""
    from deepsignal_amd import weights as W
    from oracle import oracle, torch_statement
    w = W.random_weights(seed=33, lstm_bias_std=0.1, **geom)
    feats = synth.synthetic_features(70, seed=8, **geom)
    eng = _engine(w, max_batch=128, debug=True, precision=precision, **geom)
    act, pred = eng.run(feats["kmer"], feats["means"], feats["stds"], feats["sanums"], feats["signals"])
    if precision in FP32_CLASS:
        o_act, o_pred, taps = oracle.forward(w, feats, "f32", taps=True, **geom)
        for name, ref in taps.items():
            got = eng.intermediate(name, ref.shape)
            err = float(np.abs(got - ref).max())
            assert err <= INTERMEDIATE_RTOL * max(1.0, float(np.abs(ref).max())), (name, err)
        _check_outputs(act, pred, o_act, o_pred)
    else:
        e_act, _ = torch_statement.forward_bf16(w, feats, lstm_bf16=True)
        assert np.isfinite(act).all() and np.abs(act - e_act).max() <= 3e-3
    eng.close()


@pytest.mark.parametrize("geom,batch", [(dict(kmer_len=17, signal_len=128), 1024), (dict(kmer_len=17, signal_len=40), 2560),
                                        (dict(kmer_len=17, signal_len=62), 1536)])
def test_short_signals_at_big_batches(geom, batch):
    """A non-default --cent_signals_len with a batch big enough for stem23_kernel's fullest tiles (spt = 96 // wa whole
    sites): the T tile's halo rows pushed its LDS request past the 80 KB the kernel may ask for (wa = 32 at >= 770
    sites, wa = 10 at >= 2300) and every forward failed at launch. The planner now shrinks the tile; sampled sites
    against the oracle, and the bits of a site must not depend on the batch it travels in."""
    from deepsignal_amd import weights as W
    from oracle import oracle
    w = W.random_weights(seed=35, lstm_bias_std=0.1, **geom)
    feats = synth.synthetic_features(batch, seed=9, **geom)
    keys = ("kmer", "means", "stds", "sanums", "signals")
    eng = _engine(w, max_batch=batch, **geom)
    act, pred = eng.run(*(feats[k] for k in keys))
    assert np.isfinite(act).all()
    sel = np.random.default_rng(1).choice(batch, 96, replace=False)
    o_act, o_pred = oracle.forward(w, {k: v[sel] for k, v in feats.items()}, "f32", **geom)
    _check_outputs(act[sel], pred[sel], o_act, o_pred)
    a2, p2 = eng.run(*(feats[k][sel[:70]] for k in keys))
    assert np.array_equal(a2, act[sel[:70]]) and np.array_equal(p2, pred[sel[:70]])
    eng.close()


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "bf16", "bf16_all"])
def test_a_batch_of_8192_gives_the_bits_of_batches_of_512(small_weights, precision):
    """Every tiling decision the planner takes from the batch size (sites per fused-module tile, BiLSTM tile shapes, the
    bf16 module chain that keeps its rows in LDS, grids of > 65,535 workgroups) must leave a site's bits alone: one forward
    of 8,192 sites (and a ragged one of 8,155) against the same sites in passes of 512."""
    feats = synth.synthetic_features(8192, seed=91)
    keys = ("kmer", "means", "stds", "sanums", "signals")
    ref = _engine(small_weights, max_batch=512, precision=precision)
    r_act, r_pred = ref.run(*(feats[k] for k in keys))
    ref.close()
    big = _engine(small_weights, max_batch=8192, precision=precision)
    act, pred = big.run(*(feats[k] for k in keys))
    assert np.array_equal(act, r_act) and np.array_equal(pred, r_pred)
    act, pred = big.run(*(feats[k][37:] for k in keys))
    assert np.array_equal(act, r_act[37:]) and np.array_equal(pred, r_pred[37:])
    big.close()


def test_submit_wait_matches_run(small_weights):
    """Asynchronous host boundary (ds_submit / ds_wait): tickets waited in order give the bits of the blocking run();
    over-subscription and stale tickets are refused."""
    feats = synth.synthetic_features(5 * 96 + 17, seed=41)
    keys = ("kmer", "means", "stds", "sanums", "signals")
    eng = _engine(small_weights, max_batch=96, slots=3)
    assert eng.slots == 3
    ref_act, ref_pred = eng.run(*(feats[k] for k in keys))
    chunks = [(s, min(s + 96, len(feats["kmer"]))) for s in range(0, len(feats["kmer"]), 96)]
    got_act, got_pred, inflight = [], [], []
    for s, e in chunks:
        if len(inflight) == eng.slots:
            a, p = eng.wait(inflight.pop(0))
            got_act.append(a); got_pred.append(p)
        inflight.append(eng.submit(*(feats[k][s:e] for k in keys)))
    with pytest.raises(RuntimeError):          # all three slots are in flight
        eng.submit(*(feats[k][:8] for k in keys))
    while inflight:
        a, p = eng.wait(inflight.pop(0))
        got_act.append(a); got_pred.append(p)
    assert np.array_equal(np.concatenate(got_act), ref_act) and np.array_equal(np.concatenate(got_pred), ref_pred)
    with pytest.raises(RuntimeError):          # nothing in flight any more
        eng.wait((0, 8))
    # ds_submit_parts: the same batch as ragged row segments (empty ones included) gives the same bits
    cuts = [0, 1, 1, 40, 77, 96]
    parts = [tuple(feats[k][s:e] for k in keys) for s, e in zip(cuts[:-1], cuts[1:])]
    a, p = eng.wait(eng.submit_parts(parts))
    assert np.array_equal(a, ref_act[:96]) and np.array_equal(p, ref_pred[:96])
    with pytest.raises(RuntimeError):          # more rows than max_batch
        eng.submit_parts(parts + parts)
    eng.close()


def test_config2_full_size_with_sampled_oracle_check():
    """BASELINE configs[1] / SURVEY.md 8d "Config 2" at its full size: 512 x 200 sites in memory, benchmark weights,
    fp32. The oracle (~1 k sites/s) checks a 2,048-site random sample; size-independent properties cover the rest:
    finite outputs, probabilities in (0, 1), and a site's bits do not depend on where in the stream it sits."""
    from deepsignal_amd import weights as W
    from oracle import oracle
    n = 512 * 200
    w = W.random_weights(seed=W.WEIGHT_SEED)
    feats = synth.synthetic_features(n, seed=synth.FEATURE_SEED)
    keys = ("kmer", "means", "stds", "sanums", "signals")
    eng = _engine(w, max_batch=512)
    act, pred = eng.run(*(feats[k] for k in keys))
    assert act.shape == (n, 2) and np.isfinite(act).all() and (act > 0).all() and (act < 1).all()
    assert np.array_equal(pred, np.argmax(act, axis=1).astype(np.int32))
    rng = np.random.default_rng(0)
    sel = np.sort(rng.choice(n, 2048, replace=False))
    sub = {k: v[sel] for k, v in feats.items()}
    o_act, o_pred = oracle.forward(w, sub, "f32")
    pn = _norm(act[sel]), _norm(o_act)
    assert np.abs(pn[0] - pn[1]).max() <= 1e-4            # the north-star gate; measured ~4e-7
    _check_outputs(act[sel], pred[sel], o_act, o_pred)
    # the same sites, re-run as one odd-sized batch elsewhere in the stream: identical bits
    a2, p2 = eng.run(*(feats[k][sel[:300]] for k in keys))
    assert np.array_equal(a2, act[sel[:300]]) and np.array_equal(p2, pred[sel[:300]])
    eng.close()


def test_narrow_and_wide_lstm_tilings_give_the_same_bits(small_weights):
    """The BiLSTM cells run on 128 x 32 tiles (transposed MFMA, [gate][8 units] column order) for forwards of <= 512
    sites and on 128 x 128 tiles otherwise; both accumulate K in the same order and round the gate math identically."""
    feats = synth.synthetic_features(200, seed=44)
    keys = ("kmer", "means", "stds", "sanums", "signals")
    outs = {}
    for tag in ("wide", "narrow"):
        eng = _engine(small_weights, max_batch=256, debug=True, lstm_tiling=tag)
        act, pred = eng.run(*(feats[k] for k in keys))
        outs[tag] = (act, pred, eng.intermediate("lstm_fw_l0", (200, 17, 256)), eng.intermediate("lstm_bw_l2", (200, 17, 256)))
        eng.close()
    for a, b in zip(outs["wide"], outs["narrow"]):
        assert np.array_equal(a, b)


def test_many_ragged_sizes_bounded_plan_cache(small_weights):
    """A long run over queue items sees many distinct tail sizes; the per-slot plan cache is bounded (least recently
    used sizes are dropped) and graphs are only captured for recurring sizes -- results must not depend on any of it."""
    feats = synth.synthetic_features(64, seed=61)
    keys = ("kmer", "means", "stds", "sanums", "signals")
    eng = _engine(small_weights, max_batch=64, slots=2)
    ref_act, ref_pred = eng.run(*(feats[k] for k in keys))
    for rep in range(2):
        for n in list(range(1, 41)) + [64, 17, 3, 64]:          # 40+ sizes > the cache bound, some recurring
            a, p = eng.run(*(feats[k][:n] for k in keys))
            assert np.array_equal(a, ref_act[:n]) and np.array_equal(p, ref_pred[:n]), (rep, n)
    eng.close()


def test_class_num_three():
    """--class_num is a CLI parameter of the reference (deepsignal.py:266-267); the head handles any class count."""
    from deepsignal_amd import weights as W
    from oracle import oracle
    w = W.random_weights(seed=8, lstm_bias_std=0.1, class_num=3)
    feats = synth.synthetic_features(50, seed=19)
    eng = _engine(w, max_batch=64, class_num=3)
    act, pred = eng.run(feats["kmer"], feats["means"], feats["stds"], feats["sanums"], feats["signals"])
    o_act, o_pred = oracle.forward(w, feats, "f32", class_num=3)
    assert act.shape == (50, 3) and np.abs(act - o_act).max() <= ACT_ATOL
    srt = np.sort(o_act, axis=1)
    decided = srt[:, -1] - srt[:, -2] > 1e-3
    assert (pred[decided] == o_pred[decided]).all()
    eng.close()


@pytest.mark.parametrize("variant", [dict(), dict(is_cnn=False), dict(is_rnn=False), dict(class_num=3)])
def test_folded_joint_model_matches_the_three_step_path(variant):
    """Default fp32 engine: avgpool + dense(J, J) + dense(J, C) (layers.py:233-238,257-263: no bias, no activation,
    identity dropout) are folded into ONE J x C matrix at weight load (float64 product). DS_TUNE_NO_FOLD_FC keeps the
    reference's three steps. Same function, different rounding: logits agree to fp32 rounding, both agree with the
    oracle, and the folded handle refuses the taps it no longer computes."""
    from deepsignal_amd import weights as W
    from oracle import oracle
    w = W.random_weights(seed=21, lstm_bias_std=0.1, **variant)
    feats = synth.synthetic_features(300, seed=77)
    keys = ("kmer", "means", "stds", "sanums", "signals")
    folded = _engine(w, max_batch=512, **variant)
    steps = _engine(w, max_batch=512, fold_fc=False, **variant)
    act_f, pred_f = folded.run(*[feats[k] for k in keys])
    act_s, pred_s = steps.run(*[feats[k] for k in keys])
    C = variant.get("class_num", 2)
    lf, ls = folded.intermediate("logits", (300, C)), steps.intermediate("logits", (300, C))
    assert np.abs(lf - ls).max() <= 2e-6 * max(1.0, float(np.abs(ls).max()))
    assert np.abs(act_f - act_s).max() <= 2e-6
    o_act, o_pred = oracle.forward(w, feats, "f32", **variant)
    _check_outputs(act_f, pred_f, o_act, o_pred)
    _check_outputs(act_s, pred_s, o_act, o_pred)
    with pytest.raises(RuntimeError):
        folded.intermediate("fc1", (300, steps_j(variant)))
    assert steps.intermediate("fc1", (300, steps_j(variant))).shape[1] == steps_j(variant)
    assert "fc1" in {s["name"] for s in steps.stage_times()} and "fc1" not in {s["name"] for s in folded.stage_times()}
    folded.close(); steps.close()


def steps_j(variant):
    return spec.net_dims(is_cnn=variant.get("is_cnn", True), is_rnn=variant.get("is_rnn", True)).joint


@pytest.mark.parametrize("precision", ["fp32", "bf16_all"])
def test_taps_that_do_not_exist_are_refused(small_weights, precision):
    """Outside debug mode conv_layer2's rows never leave LDS, the module buffers are shared (and in the bf16 modes a chain's
    inner modules keep their rows in LDS), and the folded joint model has no pooled features / fc1: those taps must fail
    loudly, not hand back whatever the buffer last held. The last module's rows and the logits always exist."""
    feats = synth.synthetic_features(40, seed=5)
    keys = ("kmer", "means", "stds", "sanums", "signals")
    eng = _engine(small_weights, max_batch=64, precision=precision)
    eng.run(*(feats[k] for k in keys))
    for name, shape in (("stem_conv2", (40, 90, 128)), ("module5", (40, 45, 240)), ("module10", (40, 23, 240)),
                        ("signal_feat", (40, 5520)), ("joint", (40, 6032)), ("fc1", (40, 6032))):
        with pytest.raises(RuntimeError):
            eng.intermediate(name, shape)
    assert np.isfinite(eng.intermediate("module11", (40, 23, 240))).all()
    assert np.isfinite(eng.intermediate("logits", (40, 2))).all()
    eng.close()
    dbg = _engine(small_weights, max_batch=64, precision=precision, debug=True)
    dbg.run(*(feats[k] for k in keys))
    for name, shape in (("stem_conv2", (40, 90, 128)), ("module5", (40, 45, 240)), ("module10", (40, 23, 240)), ("fc1", (40, 6032))):
        assert np.isfinite(dbg.intermediate(name, shape)).all()
    dbg.close()
