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
