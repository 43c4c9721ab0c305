"""CPU (-m "not gpu"): the oracle against the committed golden vectors, the independent torch
statement, and the properties the domain offers. Parity is UNPINNED by the reference (no TF1, no
reference tests); these are the strongest pins available."""
import os

import numpy as np
import pytest
import torch

from deepsignal_amd import spec, synth, weights
from oracle import oracle
from oracle import torch_statement

GOLD = os.path.join(os.path.dirname(__file__), "golden", "forward_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(GOLD))


@pytest.fixture(scope="module")
def gold_weights(gold):
    return weights.random_weights(seed=int(gold["weight_seed"]), lstm_bias_std=float(gold["lstm_bias_std"]))


def _feats(gold):
    return {k[3:]: gold[k] for k in gold if k.startswith("in_")}


def test_spec_counts():
    d = spec.net_dims()
    assert (d.w_conv1, d.w_a, d.w_b, d.w_c) == (180, 90, 45, 23)
    assert d.pad_conv1 == (2, 3) and d.pad_pool1 == (0, 1) and d.pad_pool2 == (0, 1) and d.pad_pool3 == (1, 1)
    assert d.joint == 6032 and spec.param_count() == 40_426_464
    assert len(spec.tensor_table()) == 580
    # FLOP contract of SURVEY.md 8(d): recompute MACs from the spec
    conv = 180 * 7 * 64 + 90 * 64 * 128 + 90 * 3 * 128 * 256
    for n in range(1, 12):
        w, cin = d.module_width(n), d.module_cin(n)
        conv += w * (cin * 240 + 96 * 48 + 160 * 48 + 96 * 64 + 64 * 48)
    assert 2 * conv == spec.FLOPS_CONV_PER_SITE
    lstm = 2 * 17 * ((131 + 256) * 1024 + 2 * 512 * 1024)
    assert 2 * lstm == spec.FLOPS_LSTM_PER_SITE
    assert 2 * (6032 * 6032 + 6032 * 2) == spec.FLOPS_FC_PER_SITE


def test_oracle_f64_matches_golden(gold, gold_weights):
    act, pred, taps = oracle.forward(gold_weights, _feats(gold), "f64", taps=True)
    assert np.array_equal(pred, gold["pred"])
    assert np.abs(act - gold["act"]).max() == 0.0
    assert np.abs(taps["logits"] - gold["logits"]).max() == 0.0
    assert np.abs(taps["module11"] - gold["module11"]).max() == 0.0
    assert np.abs(taps["stem_pool"][0] - gold["stem_pool_site0"]).max() == 0.0
    assert np.abs(taps["module4"][1] - gold["module4_site1"]).max() == 0.0
    assert np.abs(taps["lstm_fw_l2"][:, -1] - gold["lstm_fw_l2_last"]).max() == 0.0
    assert np.abs(taps["lstm_bw_l2"][:, 0] - gold["lstm_bw_l2_first"]).max() == 0.0
    assert np.abs(taps["fc1"][:, :512] - gold["fc1_head"]).max() == 0.0


def test_oracle_f32_close_to_golden(gold, gold_weights):
    act, pred, taps = oracle.forward(gold_weights, _feats(gold), "f32", taps=True)
    assert np.abs(act - gold["act"]).max() < 2e-6
    assert np.abs(taps["module11"] - gold["module11"]).max() < 2e-5
    assert np.abs(taps["fc1"][:, :512] - gold["fc1_head"]).max() < 2e-5
    decided = np.abs(gold["act"][:, 1] - gold["act"][:, 0]) > 1e-3
    assert (pred[decided] == gold["pred"][decided]).all()


def test_independent_torch_statement_agrees(gold, gold_weights):
    """Two independent statements of the TF-1.x semantics (C loops vs torch library ops)."""
    feats = _feats(gold)
    o_act, o_pred, o_taps = oracle.forward(gold_weights, feats, "f64", taps=True)
    t_act, t_pred, t_taps = torch_statement.forward(gold_weights, feats, torch.float64, True)
    assert np.abs(o_act - t_act).max() < 1e-7 and np.array_equal(o_pred, t_pred)
    for k, v in o_taps.items():
        assert np.abs(v - t_taps[k]).max() <= 1e-6 * max(1.0, np.abs(v).max()), k


def test_third_statement_in_nn_modules_agrees(gold, gold_weights):
    """oracle/nn_statement.py: torch.nn.LSTM with the TF kernel's gate blocks permuted (i,j,f,o -> i,f,g,o) and
    forget_bias folded into the bias, nn.Conv1d(padding="same"), nn.BatchNorm1d(eps=1e-3).eval(), ceil_mode max-pools,
    nn.AvgPool1d(count_include_pad=False), nn.Linear -- library code neither other statement goes through. All three
    must give the same numbers (float64, < 1 fp32 ulp on every tensor it taps) and the committed golden outputs."""
    from oracle import nn_statement
    feats = _feats(gold)
    o_act, o_pred, o_taps = oracle.forward(gold_weights, feats, "f64", taps=True)
    n_act, n_pred, n_taps = nn_statement.forward(gold_weights, feats, torch.float64, True)
    assert np.abs(o_act - n_act).max() < 1e-7 and np.array_equal(o_pred, n_pred)
    assert np.abs(n_act - gold["act"]).max() < 1e-7 and np.array_equal(n_pred, gold["pred"])
    assert set(n_taps) >= {"lstm_fw_l2", "lstm_bw_l2", "stem_pool", "module1", "module4", "module9", "module11", "fc1"}
    for k, v in n_taps.items():
        assert np.abs(v - o_taps[k]).max() <= 1e-6 * max(1.0, np.abs(o_taps[k]).max()), k
    # fp32 library kernels (oneDNN / MKL, fused LSTM cell) against the fp32 C oracle: summation order is all that differs
    f_act, f_pred = nn_statement.forward(gold_weights, feats, torch.float32)
    c_act, c_pred = oracle.forward(gold_weights, feats, "f32")
    assert np.abs(f_act - c_act).max() < 1e-5


def test_site_independence_and_batch_order(gold, gold_weights):
    """Every site is an independent forward (no cross-site state): permuting / slicing the batch
    permutes / slices the output bit-exactly."""
    feats = _feats(gold)
    act, pred = oracle.forward(gold_weights, feats, "f32")
    perm = np.random.default_rng(0).permutation(len(pred))
    p_act, p_pred = oracle.forward(gold_weights, {k: v[perm] for k, v in feats.items()}, "f32")
    assert np.array_equal(p_act, act[perm]) and np.array_equal(p_pred, pred[perm])
    one_act, _ = oracle.forward(gold_weights, {k: v[3:4] for k, v in feats.items()}, "f32")
    assert np.array_equal(one_act[0], act[3])


def test_semantics_spot_checks(gold_weights):
    """Points where TF-1.x semantics are easy to get wrong."""
    # labels / learning-rate do not influence inference; sigmoid (not softmax) head: outputs need not sum to 1
    feats = synth.synthetic_features(4, seed=5)
    act, pred = oracle.forward(gold_weights, feats, "f64")
    assert ((act > 0) & (act < 1)).all() and np.abs(act.sum(axis=1) - 1).max() > 1e-3
    assert np.array_equal(pred, np.argmax(act, axis=1))
    # a short window zero-padded on the right is just an input (extract_features.py:157-160)
    f2 = {k: v.copy() for k, v in feats.items()}
    f2["signals"][:, 100:] = 0
    a2, _ = oracle.forward(gold_weights, f2, "f64")
    assert np.isfinite(a2).all()
    # avg-pool divisor = in-bounds taps: a constant module-11 map must stay constant after pooling
    x = np.ones((23, 4))
    out = np.stack([x[max(0, w - 3):w + 4].mean(axis=0) for w in range(23)])
    assert np.allclose(out, 1.0)


def test_weight_file_roundtrip(tmp_path, gold_weights):
    sub = {k: gold_weights[k] for k in list(gold_weights)[:40]}
    p = str(tmp_path / "w.dsw")
    weights.save_weights(p, sub)
    back = weights.load_weights(p)
    assert list(back) == list(sub)
    for k in sub:
        assert np.array_equal(back[k], sub[k]) and back[k].dtype == np.float32
    with pytest.raises(KeyError):
        weights.check_weights(sub)


@pytest.mark.parametrize("variant", [dict(is_cnn=False, is_rnn=True, is_base=True),
                                     dict(is_cnn=True, is_rnn=False, is_base=True),
                                     dict(is_cnn=True, is_rnn=True, is_base=False)])
def test_model_variants_two_statements_agree(variant):
    """Model(is_cnn, is_rnn, is_base) switches (model.py:28-29,59-75,89-95): joint width, inputs of the
    first LSTM layer and the tensor table all change; both statements must still agree."""
    w = weights.random_weights(seed=5, lstm_bias_std=0.1, **variant)
    weights.check_weights(w, **variant)
    feats = synth.synthetic_features(5, seed=3)
    o_act, o_pred, o_taps = oracle.forward(w, feats, "f64", taps=True, **variant)
    t_act, t_pred, t_taps = torch_statement.forward(w, feats, torch.float64, True, **variant)
    assert set(o_taps) == set(t_taps)
    assert np.abs(o_act - t_act).max() < 1e-7 and np.array_equal(o_pred, t_pred)
    for k, v in o_taps.items():
        assert np.abs(v - t_taps[k]).max() <= 1e-6 * max(1.0, np.abs(v).max()), k
    d = spec.net_dims(is_cnn=variant["is_cnn"], is_rnn=variant["is_rnn"])
    assert o_taps["joint"].shape[1] == d.joint == (5520 if variant["is_cnn"] else 0) + (512 if variant["is_rnn"] else 0)
    with pytest.raises(ValueError):
        spec.net_dims(is_cnn=False, is_rnn=False)


def test_stress_golden_vectors_and_balanced_heads():
    """The trained-regime fixture (tests/golden/make_stress_golden.py): the float64 oracle reproduces the committed outputs, the
    fp32 oracle sits within 1e-4 of them (3 - 5e-5 measured: the noise floor tests/test_gpu_stress.py holds the HIP path to),
    both labels occur, the LSTM saturates, and both committed heads are what weights.centred_head makes of the oracle's fc1."""
    import os
    from deepsignal_amd import synth, weights as W
    from oracle import oracle
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stress_golden.npz"))
    w = W.stress_weights(int(g["stress_seed"]), head=g["stress_head"])
    feats = {k: g["in_" + k] for k in ("kmer", "means", "stds", "sanums", "signals")}
    a64, p64, t64 = oracle.forward(w, feats, "f64", taps=True)
    assert np.abs(a64 - g["act"]).max() < 1e-9 and (p64 == g["pred"]).all()
    assert np.abs(t64["logits"] - g["logits"]).max() < 1e-6
    a32, p32 = oracle.forward(w, feats, "f32")
    assert np.abs(a32 - g["act"]).max() <= 1e-4
    assert 0.25 < g["pred"].mean() < 0.75 and np.abs(g["logits"]).max() > 8.0
    assert np.abs(t64["lstm_fw_l2"]).max() > 0.9
    # the heads: regenerated from the oracle's fc1 of the probe batch. The committed heads are float32 roundings of a float64 forward
    # and reductions (BLAS matmuls, std / mean): another BLAS build or thread count may move an entry by one float32 ulp, so
    # the comparison allows that and nothing more (the purely RNG-driven parts, stress_weights, are compared bit for bit elsewhere)
    for tag, wset, seed, std in (("stress", W.stress_weights(int(g["stress_seed"])), int(g["stress_seed"]), float(g["stress_logit_std"])),
                                 ("small", W.random_weights(seed=int(g["small_seed"]), lstm_bias_std=float(g["small_lstm_bias_std"])),
                                  int(g["small_seed"]), float(g["small_logit_std"]))):
        probe = synth.synthetic_features(96, seed=seed + 2)
        _, _, taps = oracle.forward(wset, probe, "f64", taps=True)
        head = W.centred_head(taps["fc1"], wset["dense_1/kernel"][:, 0], std, seed + 3)
        ref = g[tag + "_head"]
        assert head.shape == ref.shape and np.allclose(head, ref, rtol=1e-6, atol=1e-6 * float(np.abs(ref).max())), tag


def test_split_operand_statement_is_exact_in_its_terms_and_fp32_class():
    """oracle/torch_statement.py::forward_split is the CPU statement of DS_PRECISION_BF16X3 (tests/test_gpu_split.py holds the HIP
    engine to it). Pinned here without a GPU: three bf16 terms reproduce an fp32 value exactly, the six-product sum of a matrix
    product is as close to float64 as a native fp32 product, and the whole forward with every matrix product split lands within
    the fp32 oracle's own distance of the float64 oracle (two terms / three products do not: 16 significant bits)."""
    import torch
    from deepsignal_amd import synth, weights as W
    from oracle import oracle, torch_statement as ts
    rng = np.random.default_rng(5)
    x = torch.from_numpy((rng.normal(size=4096) * np.exp(rng.uniform(-20, 20, size=4096))).astype(np.float32))
    t = ts.split_terms(x, 3)
    assert all(torch.equal(v, v.to(torch.bfloat16).to(torch.float32)) for v in t)            # every term is a bf16 value
    assert torch.equal(t[0] + t[1] + t[2], x)                                                 # ... and they sum to x exactly
    assert ts.split_pairs(3) == [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)] and ts.split_pairs(2) == [(0, 0), (0, 1), (1, 0)]
    a = torch.from_numpy(rng.normal(size=(64, 512)).astype(np.float32))
    b = torch.from_numpy(rng.normal(size=(512, 96)).astype(np.float32))
    ref = a.double() @ b.double()
    e3 = float((ts._split_matmul(a, b, 3, torch.float32).double() - ref).abs().max())
    e32 = float(((a @ b).double() - ref).abs().max())
    e2 = float((ts._split_matmul(a, b, 2, torch.float32).double() - ref).abs().max())
    assert e3 <= 2.0 * e32 + 1e-6 and e2 > 10.0 * e3, (e3, e32, e2)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stress_golden.npz"))
    w = W.stress_weights(int(g["stress_seed"]), head=g["stress_head"])
    feats = synth.synthetic_features(24, seed=77)
    a64, p64 = oracle.forward(w, feats, "f64")
    a32, _ = oracle.forward(w, feats, "f32")
    a3, p3 = ts.forward_split(w, feats, terms=3, scope=("modules", "stem23", "lstm", "fc1"))
    d3, d32 = float(np.abs(a3 - a64).max()), float(np.abs(a32 - a64).max())
    assert d3 <= max(2.0 * d32, 4e-5) and d3 <= 1e-4, (d3, d32)
    decided = np.abs(a64[:, 1] - a64[:, 0]) > 1e-3
    assert (p3[decided] == p64[decided]).all()
