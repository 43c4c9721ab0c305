"""Independent PyTorch-CPU statement of the call_mods forward pass (TEST INFRASTRUCTURE).

Written separately from oracle/ds_oracle.c, with library ops instead of loops (F.conv1d on NCW
tensors with explicit asymmetric SAME pads, F.max_pool1d over -inf padding, avg_pool1d with
count_include_pad=False, an LSTM from raw matmuls in TF gate order i,j,f,o). Two independent
statements of SURVEY.md Appendix A/B agreeing is the strongest pin available, because TensorFlow 1.x
itself cannot run here (parity unpinned by the reference).

Reference call sites: deepsignal/model.py:61-108, deepsignal/layers.py:20-264.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from deepsignal_amd import spec


def _same(x: torch.Tensor, k: int, s: int, value: float = 0.0) -> torch.Tensor:
    _, l, r = spec.same_pad(x.shape[-1], k, s)
    return F.pad(x, (l, r), value=value)


def _conv_bn(x, w: Dict[str, torch.Tensor], c: spec.ConvBN):
    # HWIO [1,K,Cin,Cout] -> torch [Cout,Cin,K]
    ker = w[c.kernel_name][0].permute(2, 1, 0).contiguous()
    y = F.conv1d(_same(x, c.k, c.stride), ker, stride=c.stride)
    g, b = w[c.bn_tensor("gamma")], w[c.bn_tensor("beta")]
    m, v = w[c.bn_tensor("moving_mean")], w[c.bn_tensor("moving_variance")]
    y = (y - m[None, :, None]) * (g / torch.sqrt(v + spec.BN_EPS))[None, :, None] + b[None, :, None]
    return torch.relu(y) if c.relu else y


def _maxpool3(x, s):
    return F.max_pool1d(_same(x, 3, s, value=float("-inf")), 3, stride=s)


def _inception(x, w, n, cin):
    c = spec.inception_convs(n, cin)
    b1 = _conv_bn(_maxpool3(x, 1), w, c["b1"])
    b2 = _conv_bn(x, w, c["b2"])
    b3 = _conv_bn(_conv_bn(x, w, c["b3a"]), w, c["b3b"])
    b4 = _conv_bn(_conv_bn(x, w, c["b4a"]), w, c["b4b"])
    stem = _conv_bn(x, w, c["b5s"])
    r = _conv_bn(_conv_bn(_conv_bn(x, w, c["b5a"]), w, c["b5b"]), w, c["b5c"])
    b5 = torch.relu(stem + r)
    return torch.cat([b1, b2, b3, b4, b5], dim=1)


def forward(weights: Dict[str, np.ndarray], feats: Dict[str, np.ndarray], dtype=torch.float64,
            return_taps: bool = False, is_cnn: bool = True, is_rnn: bool = True, is_base: bool = True):
    w = {k: torch.from_numpy(np.asarray(v)).to(dtype) for k, v in weights.items()}
    kmer = torch.from_numpy(feats["kmer"]).long()
    n, T = kmer.shape
    d = spec.net_dims(T, feats["signals"].shape[1], w["dense_1/kernel"].shape[1], is_cnn, is_rnn)
    taps = {}
    parts = []
    if is_rnn:
        parts += _event_model(w, feats, kmer, n, T, dtype, taps, is_base)
    if is_cnn:
        parts.append(_signal_model(w, feats, n, d, dtype, taps))
    joint = torch.cat(parts, dim=1)
    fc1 = joint @ w["dense/kernel"]
    logits = fc1 @ w["dense_1/kernel"]
    act = torch.sigmoid(logits)
    pred = torch.argmax(act, dim=1)
    taps.update(joint=joint, fc1=fc1, logits=logits)
    if return_taps:
        out = {}
        for k, v in taps.items():
            v = v.permute(0, 2, 1) if (k.startswith("stem") or k.startswith("module")) else v
            out[k] = v.contiguous().to(torch.float32).numpy()
        return act.to(torch.float32).numpy(), pred.to(torch.int32).numpy(), out
    return act.to(torch.float32).numpy(), pred.to(torch.int32).numpy()


def _signal_model(w, feats, n, d, dtype, taps):
    # --- signal model (NCW) ---
    x = torch.from_numpy(feats["signals"]).to(dtype)[:, None, :]
    stem = spec.stem_convs()
    x = _maxpool3(_conv_bn(x, w, stem[0]), 2)
    taps["stem_pool"] = x
    x = _conv_bn(x, w, stem[1]); taps["stem_conv2"] = x
    x = _conv_bn(x, w, stem[2]); taps["stem_conv3"] = x
    for m in range(1, spec.N_INCEPTION + 1):
        x = _inception(x, w, m, d.module_cin(m))
        taps["module%d" % m] = x
        if m in (3, 8):
            x = _maxpool3(x, 2)
    x = F.avg_pool1d(x, 7, stride=1, padding=3, count_include_pad=False)
    signal_feat = x.permute(0, 2, 1).reshape(n, -1)          # flatten order (w, c)
    taps["signal_feat"] = signal_feat
    return signal_feat


def _event_model(w, feats, kmer, n, T, dtype, taps, is_base):
    # --- event model ---
    extra = [torch.from_numpy(feats[k]).to(dtype)[:, :, None] for k in ("means", "stds", "sanums")]
    if is_base:
        emb = w[spec.MODEL_PREFIX + "embedding"][kmer]        # [n,T,128]
        x0 = torch.cat([emb] + extra, dim=2)                  # [n,T,131]
    else:
        x0 = torch.cat(extra, dim=2)                          # [n,T,3]   (model.py:70-75)
    outs = []
    for direction in ("fw", "bw"):
        seq = x0 if direction == "fw" else torch.flip(x0, dims=[1])
        for layer in range(spec.LSTM_LAYERS):
            K = w[spec.lstm_tensor(direction, layer, "kernel")]
            b = w[spec.lstm_tensor(direction, layer, "bias")]
            h = torch.zeros(n, spec.HIDDEN, dtype=dtype)
            c = torch.zeros(n, spec.HIDDEN, dtype=dtype)
            hs = []
            for t in range(T):
                z = torch.cat([seq[:, t, :], h], dim=1) @ K + b
                i, j, f, o = torch.split(z, spec.HIDDEN, dim=1)
                c = torch.sigmoid(f + spec.FORGET_BIAS) * c + torch.sigmoid(i) * torch.tanh(j)
                h = torch.sigmoid(o) * torch.tanh(c)
                hs.append(h)
            seq = torch.stack(hs, dim=1)
            taps["lstm_%s_l%d" % (direction, layer)] = seq if direction == "fw" else torch.flip(seq, dims=[1])
        outs.append(seq[:, -1, :])      # fw: t=T-1 ; bw: last processed step == original t=0
    return outs


# ---------------------------------------------------------------------------------------------
# Mixed-precision statement (BASELINE.json configs[2]; include/deepsignal_hip.h DS_PRECISION_BF16):
# bf16 operands / fp32 accumulation for the signal model's convolutions and the joint FC, activations
# between those layers stored as bf16; fp32 BiLSTM, Cin=1 stem conv, FC2, sigmoid, argmax.
# It rounds at exactly the points the engine rounds, so the HIP path can be held to a tight bound
# (differences come only from fp32 accumulation order flipping an occasional bf16 rounding).
def _rb(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.float32).to(torch.bfloat16).to(x.dtype)


def _conv_folded(x, w, c: spec.ConvBN, round_w: bool):
    """conv with BN folded into (weights, bias) in float64, like ds_engine.cpp fold_conv; bf16-rounded weights."""
    ker = w[c.kernel_name][0].to(torch.float64)                      # [K, Cin, Cout]
    g, b = w[c.bn_tensor("gamma")].to(torch.float64), w[c.bn_tensor("beta")].to(torch.float64)
    m, v = w[c.bn_tensor("moving_mean")].to(torch.float64), w[c.bn_tensor("moving_variance")].to(torch.float64)
    sc = g / torch.sqrt(v + spec.BN_EPS)
    kf = (ker * sc[None, None, :]).to(torch.float32)
    bf = (b - m * sc).to(torch.float32)
    if round_w:
        kf = kf.to(torch.bfloat16).to(torch.float32)
    kt = kf.to(x.dtype).permute(2, 1, 0).contiguous()
    return F.conv1d(_same(x, c.k, c.stride), kt, stride=c.stride) + bf.to(x.dtype)[None, :, None]


def _event_model_bf16(w, feats, kmer, n, T, dtype, taps):
    """BiLSTM of DS_PRECISION_BF16_ALL: the recurrent / lower-layer h operands and the weights they meet are bf16,
    the layer-0 input projection ([embedding, mean, std, len] @ K[:131]) is full precision, accumulation, gates and
    the cell state are full precision, and every h is stored (hence consumed downstream) as bf16."""
    extra = [torch.from_numpy(feats[k]).to(dtype)[:, :, None] for k in ("means", "stds", "sanums")]
    x0 = torch.cat([w[spec.MODEL_PREFIX + "embedding"][kmer]] + extra, dim=2)
    outs = []
    for direction in ("fw", "bw"):
        seq = x0 if direction == "fw" else torch.flip(x0, dims=[1])
        for layer in range(spec.LSTM_LAYERS):
            K = w[spec.lstm_tensor(direction, layer, "kernel")]
            b = w[spec.lstm_tensor(direction, layer, "bias")]
            nin = K.shape[0] - spec.HIDDEN
            Kx = K[:nin] if layer == 0 else _rb(K[:nin])
            Kh = _rb(K[nin:])
            h = torch.zeros(n, spec.HIDDEN, dtype=dtype)
            c = torch.zeros(n, spec.HIDDEN, dtype=dtype)
            hs = []
            for t in range(T):
                z = seq[:, t, :] @ Kx + h @ Kh + b
                i, j, f, o = torch.split(z, spec.HIDDEN, dim=1)
                c = torch.sigmoid(f + spec.FORGET_BIAS) * c + torch.sigmoid(i) * torch.tanh(j)
                h = _rb(torch.sigmoid(o) * torch.tanh(c))
                hs.append(h)
            seq = torch.stack(hs, dim=1)
            taps["lstm_%s_l%d" % (direction, layer)] = seq if direction == "fw" else torch.flip(seq, dims=[1])
        outs.append(seq[:, -1, :])
    return outs


def forward_bf16(weights: Dict[str, np.ndarray], feats: Dict[str, np.ndarray], return_taps: bool = False,
                 lstm_bf16: bool = False):
    dtype = torch.float64
    w = {k: torch.from_numpy(np.asarray(v)).to(torch.float32) for k, v in weights.items()}
    w64 = {k: v.to(dtype) for k, v in w.items()}
    kmer = torch.from_numpy(feats["kmer"]).long()
    n, T = kmer.shape
    d = spec.net_dims(T, feats["signals"].shape[1], w["dense_1/kernel"].shape[1])
    taps = {}
    if lstm_bf16:
        ev = _event_model_bf16(w64, feats, kmer, n, T, dtype, taps)
    else:
        ev = _event_model(w64, feats, kmer, n, T, dtype, taps, True)     # fp32 BiLSTM in the engine; float64 here
    # signal model
    x = torch.from_numpy(feats["signals"]).to(dtype)[:, None, :]
    stem = spec.stem_convs()
    x = _rb(_maxpool3(torch.relu(_conv_folded(x, w, stem[0], False)), 2)); taps["stem_pool"] = x
    x = _rb(torch.relu(_conv_folded(x, w, stem[1], True))); taps["stem_conv2"] = x
    x = _rb(torch.relu(_conv_folded(x, w, stem[2], True))); taps["stem_conv3"] = x
    for mth in range(1, spec.N_INCEPTION + 1):
        c = spec.inception_convs(mth, d.module_cin(mth))
        cv = lambda inp, key: _conv_folded(inp, w, c[key], True)
        b1 = _rb(torch.relu(cv(_maxpool3(x, 1), "b1")))
        b2 = _rb(torch.relu(cv(x, "b2")))
        b3 = _rb(torch.relu(cv(_rb(torch.relu(cv(x, "b3a"))), "b3b")))
        b4 = _rb(torch.relu(cv(_rb(torch.relu(cv(x, "b4a"))), "b4b")))
        stem_r = cv(x, "b5s").to(torch.float32).to(dtype)              # kept fp32 by the engine
        r = cv(_rb(torch.relu(cv(_rb(torch.relu(cv(x, "b5a"))), "b5b"))), "b5c")
        b5 = _rb(torch.relu(stem_r + r))
        x = torch.cat([b1, b2, b3, b4, b5], dim=1)
        taps["module%d" % mth] = x
        if mth in (3, 8):
            x = _maxpool3(x, 2)
    x = _rb(F.avg_pool1d(x, 7, stride=1, padding=3, count_include_pad=False))
    signal_feat = x.permute(0, 2, 1).reshape(n, -1)
    taps["signal_feat"] = signal_feat
    joint = torch.cat([_rb(ev[0]), _rb(ev[1]), signal_feat], dim=1)
    fc1 = (joint @ _rb(w64["dense/kernel"])).to(torch.float32).to(dtype)
    logits = fc1 @ w64["dense_1/kernel"]
    act = torch.sigmoid(logits)
    pred = torch.argmax(act, dim=1)
    taps.update(joint=joint, fc1=fc1, logits=logits)
    if return_taps:
        out = {}
        for k, v in taps.items():
            v = v.permute(0, 2, 1) if (k.startswith("stem") or k.startswith("module")) else v
            out[k] = v.contiguous().to(torch.float32).numpy()
        return act.to(torch.float32).numpy(), pred.to(torch.int32).numpy(), out
    return act.to(torch.float32).numpy(), pred.to(torch.int32).numpy()


# ---------------------------------------------------------------------------------------------
# Split-operand statement (include/deepsignal_hip.h DS_PRECISION_SPLIT3 / DS_PRECISION_SPLIT2):
# fp32-class arithmetic on the bf16 matrix pipe. An fp32 operand x is carried as bf16 TERMS
#     x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)          (round to nearest even; the differences are exact in fp32)
# -- three terms hold all 24 significant bits of x (exactly, away from the subnormal range), two terms hold 16 -- and a
# product a*b is the fp32-accumulated sum of bf16 x bf16 products (each exact in fp32) over the term pairs (i, j) with
# i + j < terms: six products for three terms (a0b0, a0b1, a1b0, a0b2, a1b1, a2b0; the dropped ones are below 2^-24 of the
# product), three for two terms. Everything between the matrix products (bias, ReLU, residual add, pools, gates) is fp32
# and the stored activations are fp32, as in the fp32 engine; BN is folded into the weights in float64 and rounded to fp32
# once, as ds_engine.cpp does. `scope` names what runs split: "modules" (the eleven inception modules, layers.py:87-139),
# "stem23" (conv_layer2 / 3, layers.py:192-203), "lstm" (the recurrent and lower-layer products of the cells,
# layers.py:45-72), "fc1" (dense 6032 x 6032, layers.py:257-259); the rest is evaluated in `dtype`.
def split_terms(x: torch.Tensor, terms: int):
    """The bf16 terms of an fp32 tensor, each returned as float32 (values exactly representable in bf16)."""
    x = x.to(torch.float32)
    out, r = [], x
    for _ in range(terms):
        t = r.to(torch.bfloat16).to(torch.float32)
        out.append(t)
        r = r - t                      # exact in fp32: t agrees with r in its leading bits
    return out


def split_pairs(terms: int):
    return [(i, j) for s in range(terms) for i in range(s + 1) for j in (s - i,) if i < terms and j < terms]


def _split_conv(x, kf, bias, c: spec.ConvBN, terms: int, acc_dtype):
    """x [n, Cin, W] fp32-valued; kf [K, Cin, Cout] fp32 (BN folded). Sum of the term-pair convolutions, accumulated in
    `acc_dtype` (float32 = the engine's accumulator width; float64 isolates the error of the dropped products)."""
    xt = [t.to(acc_dtype) for t in split_terms(x, terms)]
    kt = [t.to(acc_dtype).permute(2, 1, 0).contiguous() for t in split_terms(kf, terms)]
    y = None
    for i, j in reversed(split_pairs(terms)):            # small products first
        p = F.conv1d(_same(xt[i], c.k, c.stride), kt[j], stride=c.stride)
        y = p if y is None else y + p
    return (y + bias.to(acc_dtype)[None, :, None]).to(torch.float32)


def _fold(w, c: spec.ConvBN):
    ker = w[c.kernel_name][0].to(torch.float64)
    g, b = w[c.bn_tensor("gamma")].to(torch.float64), w[c.bn_tensor("beta")].to(torch.float64)
    m, v = w[c.bn_tensor("moving_mean")].to(torch.float64), w[c.bn_tensor("moving_variance")].to(torch.float64)
    sc = g / torch.sqrt(v + spec.BN_EPS)
    return (ker * sc[None, None, :]).to(torch.float32), (b - m * sc).to(torch.float32)


def _split_matmul(a, b, terms, acc_dtype):
    at = [t.to(acc_dtype) for t in split_terms(a, terms)]
    bt = [t.to(acc_dtype) for t in split_terms(b, terms)]
    y = None
    for i, j in reversed(split_pairs(terms)):
        p = at[i] @ bt[j]
        y = p if y is None else y + p
    return y.to(torch.float32)


def forward_split(weights: Dict[str, np.ndarray], feats: Dict[str, np.ndarray], terms: int = 3,
                  scope=("modules",), acc_dtype=torch.float32, return_taps: bool = False):
    f32 = torch.float32
    w = {k: torch.from_numpy(np.asarray(v)).to(f32) for k, v in weights.items()}
    kmer = torch.from_numpy(feats["kmer"]).long()
    n, T = kmer.shape
    d = spec.net_dims(T, feats["signals"].shape[1], w["dense_1/kernel"].shape[1])
    taps = {}

    def conv(x, c, split):
        kf, bf = _fold(w, c)
        if split:
            return _split_conv(x, kf, bf, c, terms, acc_dtype)
        kt = kf.to(acc_dtype).permute(2, 1, 0).contiguous()
        return (F.conv1d(_same(x.to(acc_dtype), c.k, c.stride), kt, stride=c.stride) + bf.to(acc_dtype)[None, :, None]).to(f32)

    # --- event model
    extra = [torch.from_numpy(feats[k]).to(f32)[:, :, None] for k in ("means", "stds", "sanums")]
    x0 = torch.cat([w[spec.MODEL_PREFIX + "embedding"][kmer]] + extra, dim=2)
    ev = []
    for direction in ("fw", "bw"):
        seq = x0 if direction == "fw" else torch.flip(x0, dims=[1])
        for layer in range(spec.LSTM_LAYERS):
            K = w[spec.lstm_tensor(direction, layer, "kernel")]
            b = w[spec.lstm_tensor(direction, layer, "bias")]
            nin = K.shape[0] - spec.HIDDEN
            h = torch.zeros(n, spec.HIDDEN, dtype=f32)
            c = torch.zeros(n, spec.HIDDEN, dtype=f32)
            hs = []
            for t in range(T):
                xin = seq[:, t, :]
                if "lstm" in scope:
                    zx = (xin.to(acc_dtype) @ K[:nin].to(acc_dtype)).to(f32) if layer == 0 else _split_matmul(xin, K[:nin], terms, acc_dtype)
                    z = zx + _split_matmul(h, K[nin:], terms, acc_dtype) + b
                else:
                    z = (torch.cat([xin, h], dim=1).to(acc_dtype) @ K.to(acc_dtype)).to(f32) + b
                i, j, f, o = torch.split(z, spec.HIDDEN, dim=1)
                c = torch.sigmoid(f + spec.FORGET_BIAS) * c + torch.sigmoid(i) * torch.tanh(j)
                h = torch.sigmoid(o) * torch.tanh(c)
                hs.append(h)
            seq = torch.stack(hs, dim=1)
            taps["lstm_%s_l%d" % (direction, layer)] = seq if direction == "fw" else torch.flip(seq, dims=[1])
        ev.append(seq[:, -1, :])

    # --- signal model
    x = torch.from_numpy(feats["signals"]).to(f32)[:, None, :]
    stem = spec.stem_convs()
    x = _maxpool3(torch.relu(conv(x, stem[0], False)), 2); taps["stem_pool"] = x
    x = torch.relu(conv(x, stem[1], "stem23" in scope)); taps["stem_conv2"] = x
    x = torch.relu(conv(x, stem[2], "stem23" in scope)); taps["stem_conv3"] = x
    sm = "modules" in scope
    for mth in range(1, spec.N_INCEPTION + 1):
        c = spec.inception_convs(mth, d.module_cin(mth))
        b1 = torch.relu(conv(_maxpool3(x, 1), c["b1"], sm))
        b2 = torch.relu(conv(x, c["b2"], sm))
        b3 = torch.relu(conv(torch.relu(conv(x, c["b3a"], sm)), c["b3b"], sm))
        b4 = torch.relu(conv(torch.relu(conv(x, c["b4a"], sm)), c["b4b"], sm))
        r = conv(torch.relu(conv(torch.relu(conv(x, c["b5a"], sm)), c["b5b"], sm)), c["b5c"], sm)
        b5 = torch.relu(conv(x, c["b5s"], sm) + r)
        x = torch.cat([b1, b2, b3, b4, b5], dim=1)
        taps["module%d" % mth] = x
        if mth in (3, 8):
            x = _maxpool3(x, 2)
    x = F.avg_pool1d(x, 7, stride=1, padding=3, count_include_pad=False)
    signal_feat = x.permute(0, 2, 1).reshape(n, -1)
    taps["signal_feat"] = signal_feat
    joint = torch.cat([ev[0], ev[1], signal_feat], dim=1)
    if "fc1" in scope:
        fc1 = _split_matmul(joint, w["dense/kernel"], terms, acc_dtype)
    else:
        fc1 = (joint.to(acc_dtype) @ w["dense/kernel"].to(acc_dtype)).to(f32)
    logits = (fc1.to(acc_dtype) @ w["dense_1/kernel"].to(acc_dtype)).to(f32)
    act = torch.sigmoid(logits)
    pred = torch.argmax(act, dim=1)
    taps.update(joint=joint, fc1=fc1, logits=logits)
    if return_taps:
        out = {}
        for k, v in taps.items():
            v = v.permute(0, 2, 1) if (k.startswith("stem") or k.startswith("module")) else v
            out[k] = v.contiguous().to(torch.float32).numpy()
        return act.numpy(), pred.to(torch.int32).numpy(), out
    return act.numpy(), pred.to(torch.int32).numpy()
