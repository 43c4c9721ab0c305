"""DS_PRECISION_BF16X3 on the GPU: fp32 operands carried as three bf16 terms through the bf16 matrix pipe (six products per
MAC, fp32 accumulate; deepsignal_amd/csrc/ds_split.hip). Reference arithmetic: layers.py:87-139 (inception_layer),
layers.py:205-232 (the eleven modules).

The mode is held to the bars of native fp32 -- tests/test_gpu_parity.py and tests/test_gpu_stress.py run their layer-wise,
ragged-batch, edge-input, other-geometry and trained-regime checks once per precision of FP32_CLASS with the tolerances
unchanged. This file adds what is specific to the mode:

  * exactness against the CPU statement of the SAME arithmetic (oracle/torch_statement.py::forward_split rounds where the
    engine rounds): only the fp32 summation order differs, so the bar is tighter than the fp32-oracle bar;
  * the split kernels really ran (a silent fall-back to the native fp32 kernels would pass every parity test);
  * chained launch == one launch per module, bit for bit;
  * configurations the mode does not implement are refused loudly.
"""
import numpy as np
import pytest

from deepsignal_amd import synth

pytestmark = pytest.mark.gpu
KEYS = ("kmer", "means", "stds", "sanums", "signals")
SPLIT_SCOPE = ("modules", "stem23", "lstm", "fc1")      # what DS_PRECISION_BF16X3 runs split (ds_version() / DESIGN.md section 11)
STATEMENT_RTOL = 6e-6                 # engine vs the CPU statement of the same arithmetic, relative to the tensor's scale


def _engine(weights, **kw):
    from deepsignal_amd.engine import Engine
    eng = Engine(**kw)
    eng.load_weights(weights)
    return eng


@pytest.mark.parametrize("which", ["small", "stress"])
def test_split_engine_against_the_cpu_statement_of_the_same_arithmetic(request, which):
    from oracle import torch_statement
    w = request.getfixturevalue(which + "_weights")
    n = 96
    feats = synth.synthetic_features(n, seed=4100)
    eng = _engine(w, max_batch=128, debug=True, precision="bf16x3", split_dense_min_n=1)
    act, pred = eng.run(*(feats[k] for k in KEYS))
    s_act, s_pred, taps = torch_statement.forward_split(w, feats, terms=3, scope=SPLIT_SCOPE, return_taps=True)
    worst = {}
    for name, ref in taps.items():
        if not (name.startswith("module") or name.startswith("lstm") or name in ("stem_conv2", "stem_conv3", "signal_feat", "fc1")):
            continue
        got = eng.intermediate(name, ref.shape)
        scale = max(1.0, float(np.abs(ref).max()))
        # (a saturating recurrent net amplifies the last-bit differences of the gate non-linearities -- v_exp_f32 / v_rcp_f32 here, libm
        # there: the LSTM taps of the stress set get the slack the fp32 engine needs against the fp32 oracle on the same tensors)
        worst[name] = float(np.abs(got - ref).max()) / scale / (4.0 if which == "stress" and name.startswith("lstm") else 1.0)
    print("\n%s: engine vs forward_split, relative to the tensor's scale: %s" % (which, {k: "%.1e" % v for k, v in worst.items()}))
    bad = {k: v for k, v in worst.items() if not v <= STATEMENT_RTOL}
    assert not bad, bad
    assert np.abs(act - s_act).max() <= (1e-5 if which == "small" else 1e-4)
    eng.close()


def test_the_split_kernels_are_the_ones_that_run(small_weights):
    feats = synth.synthetic_features(96, seed=4101)
    # (split_dense_min_n=1 = the default since the 128 x 96 tile: every matrix product of the three-step path runs split)
    eng = _engine(small_weights, max_batch=96, precision="bf16x3", fold_fc=False, split_dense_min_n=1)
    eng.set_graph(False)
    eng.set_profiling(1)
    eng.run(*(feats[k] for k in KEYS))
    ran = {k["name"]: k["launches"] for k in eng.kernel_stats() if k["launches"]}
    eng.close()
    for prefix in ("stem23_split_kernel", "inception_fused_split_kernel", "lstm_cell_split_kernel", "dense_split_kernel"):
        assert any(name.startswith(prefix) for name in ran), (prefix, ran)
    # ... and their native-fp32 / bf16 counterparts did not (the 6032 x 6032 GEMM template, the fp32 / bf16 cells, the fp32 / bf16 chains)
    assert not any(name.startswith(("stem23_kernel", "inception_fused_kernel", "inception_fused_bf16", "lstm_cell_lds_kernel", "lstm_cell_kernel",
                                    "lstm_cell_bf16_kernel", "gemm_kernel<1,3,4,1")) for name in ran), ran


def test_split_chain_gives_the_bits_of_one_launch_per_module(small_weights):
    feats = synth.synthetic_features(1333, seed=812)
    args = [feats[k] for k in KEYS]
    chained = _engine(small_weights, max_batch=1333, slots=1, precision="bf16x3")
    a1, p1 = chained.run(*args)
    a1s, p1s = chained.run(*(a[:77] for a in args))
    chained.close()
    single = _engine(small_weights, max_batch=1333, slots=1, precision="bf16x3", chain_modules=False)
    a2, p2 = single.run(*args)
    single.close()
    assert np.array_equal(a1, a2) and np.array_equal(p1, p2)
    assert np.array_equal(a1s, a1[:77]) and np.array_equal(p1s, p1[:77])


def test_split_lstm_tile_shapes_give_the_same_bits(small_weights):
    """lstm_cell_split_kernel runs 64 x 64, 64 x 128 or 128 x 128 workgroup tiles, the last by four or by eight waves: every shape accumulates a
    unit's K in the same order, so a site's bits do not depend on the tile -- also with an odd number of 32-site m-tiles."""
    feats = synth.synthetic_features(1100, seed=906)
    args = [feats[k] for k in KEYS]
    outs = []
    for tiling in ("narrow", "lds1", "wide", "wide8"):
        eng = _engine(small_weights, max_batch=1100, slots=1, precision="bf16x3", lstm_tiling=tiling)
        outs.append(eng.run(*args))
        eng.close()
    for a, p in outs[1:]:
        assert np.array_equal(a, outs[0][0]) and np.array_equal(p, outs[0][1])


@pytest.mark.parametrize("variant", [dict(), dict(is_base=False), dict(debug=True)])
def test_lstm_xproj_gives_the_bits_of_the_cells_own_initial_values(small_weights, variant):
    """lstm_xproj_kernel writes layer 0's accumulator-initial values of all steps (and finishes the first step's cells) in one launch
    in front of the diagonals; a cell then loads what it used to compute (lstm_acc_init, the same code in both places): the same bits,
    also with ragged sizes (a padded last m-tile), sub-batches, without the k-mer embedding, and tap by tap in debug mode."""
    kw = {k: v for k, v in variant.items() if k != "debug"}
    from deepsignal_amd import weights as W
    w = small_weights if not kw else W.random_weights(seed=21, lstm_bias_std=0.1, **kw)
    feats = synth.synthetic_features(1100, seed=907)
    args = [feats[k] for k in KEYS]
    outs = []
    for xp in (True, "all", False):
        eng = _engine(w, max_batch=1100, slots=1, precision="bf16x3", lstm_xproj=xp, debug=bool(variant.get("debug")), **kw)
        eng.set_profiling(1)
        res = [eng.run(*args), eng.run(*(a[:77] for a in args)), eng.run(*(a[:1] for a in args))]
        ran = {k["name"]: k["launches"] for k in eng.kernel_stats() if k["launches"]}
        assert ("lstm_xproj_kernel" in ran) == bool(xp), ran
        if variant.get("debug"):
            eng.run(*args)
            res.append(tuple(eng.intermediate("lstm_%s_l%d" % (d, l), (1100, 17, 256)) for d in ("fw", "bw") for l in range(3)))
        outs.append(res)
        eng.close()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_split_dense_in_ranges_of_k_at_every_forward_size(small_weights):
    """dense(J, J) of the three-step joint model runs a 256 x 192 tile with K in 4 / 2 / 1 ranges by the ENGINE's forward size (64 / 128 /
    256 tiles); head_kernel adds the partial products while it reads its row and the fc1 tap adds them on the host. Every size (ragged
    m-blocks, one m-tile) against the native fp32 engine -- the dense is linear in its inputs, so the fp32 bars hold -- and a site's
    bits do not depend on how many sites share its forward (the ranges follow max_batch, not n)."""
    feats = synth.synthetic_features(2048, seed=4107)
    for n_max, sizes in ((2048, (2048, 1100, 33)), (1024, (1024, 513, 300)), (512, (512, 300, 32, 5))):
        ref = _engine(small_weights, max_batch=n_max, slots=1, precision="fp32", fold_fc=False)
        eng = _engine(small_weights, max_batch=n_max, slots=1, precision="bf16x3", fold_fc=False)
        full = None
        for n in sizes:
            args = [feats[k][:n] for k in KEYS]
            a0, p0 = ref.run(*args)
            a1, p1 = eng.run(*args)
            f0 = ref.intermediate("fc1", (n, 6032))
            f1 = eng.intermediate("fc1", (n, 6032))
            assert np.abs(f1 - f0).max() <= 2e-5 * max(1.0, float(np.abs(f0).max())), (n_max, n, float(np.abs(f1 - f0).max()))
            assert np.abs(a1 - a0).max() <= 2e-5 and np.array_equal(p0, p1), (n_max, n, float(np.abs(a1 - a0).max()))
            if full is None:
                full = (a1, f1)
            else:
                assert np.array_equal(a1, full[0][:n]) and np.array_equal(f1, full[1][:n]), (n_max, n)
        if n_max == 512:      # DS_TUNE_SPLIT_DENSE_NARROW (timing diagnostic): the 128 x 96 tile with K in one range, same bars
            nar = _engine(small_weights, max_batch=n_max, slots=1, precision="bf16x3", fold_fc=False, split_dense_narrow=True)
            a2, p2 = nar.run(*[feats[k][:300] for k in KEYS])
            a0, p0 = ref.run(*[feats[k][:300] for k in KEYS])
            assert np.abs(a2 - a0).max() <= 2e-5 and np.array_equal(p0, p2)
            nar.close()
        ref.close(); eng.close()


def test_split_mode_refuses_what_it_does_not_implement(small_weights):
    from deepsignal_amd.engine import Engine
    with pytest.raises(RuntimeError, match="BF16X3"):
        Engine(max_batch=64, precision="bf16x3", no_fused=True)
