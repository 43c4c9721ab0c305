"""Third statement of the call_mods forward pass, in torch.nn LIBRARY MODULES (TEST INFRASTRUCTURE).

oracle/ds_oracle.c (loops) and oracle/torch_statement.py (functional ops, an LSTM written out from raw matmuls) were
both written from one reading of the TensorFlow-1.x semantics in SURVEY.md Appendix B; where they share a line of
reasoning they can share a mistake. This file goes through code neither of them touches:

  * the BiLSTM is `torch.nn.LSTM` (cuDNN-style fused cell, gate order i, f, g, o, two bias vectors). The TF LSTMCell
    kernel [in + 256, 4 * 256] with columns i, j, f, o (layers.py:49-50; Appendix B.1) is re-packed into
    weight_ih / weight_hh with the column blocks PERMUTED to PyTorch's order, and `forget_bias = 1.0` -- which TF adds
    at run time -- is FOLDED into bias_ih's forget block. If the i,j,f,o reading or the forget-bias handling of the
    other two statements were inconsistent with an ordinary LSTM, this one would disagree with them.
    TF's bidirectional_dynamic_rnn over two MultiRNNCell stacks (layers.py:66-71) = two independent 3-layer
    unidirectional nn.LSTMs, the backward one fed the time-reversed sequence (NOT nn.LSTM(bidirectional=True), which
    concatenates the directions between layers);
  * convolutions are `nn.Conv1d(padding="same")` wherever TF's SAME padding is symmetric (every stride-1 conv), batch
    norm is `nn.BatchNorm1d(eps=1e-3).eval()` with the moving statistics loaded, stride-2 SAME max-pools with the
    (0, 1) pad are `nn.MaxPool1d(3, 2, ceil_mode=True)` (the window that hangs over the right edge) and the (1, 1)
    one `padding=1`, the average pool is `nn.AvgPool1d(count_include_pad=False)`, the dense layers `nn.Linear`.

Reference call sites: deepsignal/model.py:61-108, deepsignal/layers.py:20-264. Only tests/ import this module.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from deepsignal_amd import spec


def _conv_bn(w: Dict[str, np.ndarray], c: spec.ConvBN, dtype) -> nn.Sequential:
    ker = torch.from_numpy(np.asarray(w[c.kernel_name]))[0]             # HWIO[0] = [K, Cin, Cout]
    k, cin, cout = ker.shape
    layers = []
    if c.stride == 1:
        conv = nn.Conv1d(cin, cout, k, stride=1, padding="same", bias=False)
    else:
        conv = nn.Conv1d(cin, cout, k, stride=c.stride, padding=0, bias=False)   # the caller applies the asymmetric SAME pad
    conv.weight.data = ker.permute(2, 1, 0).contiguous().to(dtype)
    bn = nn.BatchNorm1d(cout, eps=spec.BN_EPS)
    bn.weight.data = torch.from_numpy(np.asarray(w[c.bn_tensor("gamma")])).to(dtype)
    bn.bias.data = torch.from_numpy(np.asarray(w[c.bn_tensor("beta")])).to(dtype)
    bn.running_mean = torch.from_numpy(np.asarray(w[c.bn_tensor("moving_mean")])).to(dtype)
    bn.running_var = torch.from_numpy(np.asarray(w[c.bn_tensor("moving_variance")])).to(dtype)
    layers += [conv, bn]
    if c.relu:
        layers.append(nn.ReLU())
    return nn.Sequential(*layers).eval()


class _Inception(nn.Module):
    def __init__(self, w, n, cin, dtype):
        super().__init__()
        c = spec.inception_convs(n, cin)
        self.pool = nn.MaxPool1d(3, stride=1, padding=1)                   # SAME (1,1); torch pads max-pools with -inf
        self.b1 = _conv_bn(w, c["b1"], dtype)
        self.b2 = _conv_bn(w, c["b2"], dtype)
        self.b3 = nn.Sequential(_conv_bn(w, c["b3a"], dtype), _conv_bn(w, c["b3b"], dtype))
        self.b4 = nn.Sequential(_conv_bn(w, c["b4a"], dtype), _conv_bn(w, c["b4b"], dtype))
        self.stem = _conv_bn(w, c["b5s"], dtype)
        self.res = nn.Sequential(_conv_bn(w, c["b5a"], dtype), _conv_bn(w, c["b5b"], dtype), _conv_bn(w, c["b5c"], dtype))

    def forward(self, x):
        return torch.cat([self.b1(self.pool(x)), self.b2(x), self.b3(x), self.b4(x), torch.relu(self.stem(x) + self.res(x))],
                         dim=1)


def _pool_s2(width: int) -> nn.Module:
    """max_pooling2d([1,3], strides=2, SAME) for an input of `width` samples (layers.py:189-191,211-213,224-226)."""
    _, l, r = spec.same_pad(width, 3, 2)
    if (l, r) == (0, 1):
        return nn.MaxPool1d(3, stride=2, ceil_mode=True)      # last window hangs over the right edge = one padded tap
    if (l, r) == (1, 1):
        return nn.MaxPool1d(3, stride=2, padding=1)
    if (l, r) == (0, 0):
        return nn.MaxPool1d(3, stride=2)
    raise NotImplementedError("SAME pad (%d, %d) has no nn.MaxPool1d form" % (l, r))


def _lstm_stack(w, direction: str, in0: int, dtype) -> nn.LSTM:
    """One MultiRNNCell stack (layers.py:45-68) as a 3-layer unidirectional nn.LSTM."""
    H = spec.HIDDEN
    lstm = nn.LSTM(in0, H, num_layers=spec.LSTM_LAYERS, batch_first=True, bidirectional=False)
    perm = np.concatenate([np.arange(0, H), np.arange(2 * H, 3 * H), np.arange(H, 2 * H), np.arange(3 * H, 4 * H)])
    # torch gate blocks (i, f, g, o)  <-  TF column blocks (i, f, j, o) = columns [0:H], [2H:3H], [H:2H], [3H:4H]
    for layer in range(spec.LSTM_LAYERS):
        K = np.asarray(w[spec.lstm_tensor(direction, layer, "kernel")], np.float64)      # [in + H, 4H], rows: input first
        b = np.asarray(w[spec.lstm_tensor(direction, layer, "bias")], np.float64).copy()
        nin = K.shape[0] - H
        b[2 * H:3 * H] += spec.FORGET_BIAS                                                # TF adds it at run time
        getattr(lstm, "weight_ih_l%d" % layer).data = torch.from_numpy(K[:nin, perm].T.copy()).to(dtype)
        getattr(lstm, "weight_hh_l%d" % layer).data = torch.from_numpy(K[nin:, perm].T.copy()).to(dtype)
        getattr(lstm, "bias_ih_l%d" % layer).data = torch.from_numpy(b[perm]).to(dtype)
        getattr(lstm, "bias_hh_l%d" % layer).data = torch.zeros(4 * H, dtype=dtype)
    return lstm.to(dtype).eval()


@torch.no_grad()
def forward(weights: Dict[str, np.ndarray], feats: Dict[str, np.ndarray], dtype=torch.float64, return_taps: bool = False):
    """(activation_logits float32[n, C], prediction int32[n][, taps]) for the full model (is_cnn = is_rnn = is_base)."""
    w = weights
    kmer = torch.from_numpy(feats["kmer"]).long()
    n, T = kmer.shape
    S = feats["signals"].shape[1]
    d = spec.net_dims(T, S, np.asarray(w["dense_1/kernel"]).shape[1], True, True)
    taps = {}
    # ---- event model (model.py:61-69, layers.py:161-173)
    emb = F.embedding(kmer, torch.from_numpy(np.asarray(w[spec.MODEL_PREFIX + "embedding"])).to(dtype))
    x0 = torch.cat([emb] + [torch.from_numpy(feats[k]).to(dtype)[:, :, None] for k in ("means", "stds", "sanums")], dim=2)
    ev = []
    for direction in ("fw", "bw"):
        seq = x0 if direction == "fw" else torch.flip(x0, dims=[1])
        out, _ = _lstm_stack(w, direction, x0.shape[2], dtype)(seq)        # top layer's h for every step
        taps["lstm_%s_l%d" % (direction, spec.LSTM_LAYERS - 1)] = out if direction == "fw" else torch.flip(out, dims=[1])
        ev.append(out[:, -1, :])
    # ---- signal model (layers.py:181-239), NCW
    x = torch.from_numpy(feats["signals"]).to(dtype)[:, None, :]
    stem = spec.stem_convs()
    _, l, r = spec.same_pad(S, stem[0].k, stem[0].stride)
    x = _conv_bn(w, stem[0], dtype)(F.pad(x, (l, r)))                       # the one strided conv: explicit (2, 3) pad
    x = _pool_s2(x.shape[-1])(x)
    taps["stem_pool"] = x
    x = _conv_bn(w, stem[1], dtype)(x)
    x = _conv_bn(w, stem[2], dtype)(x)
    taps["stem_conv3"] = x
    for m in range(1, spec.N_INCEPTION + 1):
        x = _Inception(w, m, d.module_cin(m), dtype).eval()(x)
        taps["module%d" % m] = x
        if m in (3, 8):
            x = _pool_s2(x.shape[-1])(x)
    x = nn.AvgPool1d(7, stride=1, padding=3, count_include_pad=False)(x)
    sig = x.permute(0, 2, 1).reshape(n, -1)
    taps["signal_feat"] = sig
    # ---- joint model (layers.py:247-264) and head (model.py:100,108)
    fc1 = nn.Linear(d.joint, d.joint, bias=False)
    fc1.weight.data = torch.from_numpy(np.asarray(w["dense/kernel"])).to(dtype).t().contiguous()
    fc2 = nn.Linear(d.joint, np.asarray(w["dense_1/kernel"]).shape[1], bias=False)
    fc2.weight.data = torch.from_numpy(np.asarray(w["dense_1/kernel"])).to(dtype).t().contiguous()
    joint = torch.cat(ev + [sig], dim=1)
    h1 = fc1(joint)
    logits = fc2(h1)
    act = torch.sigmoid(logits)
    pred = torch.argmax(act, dim=1)
    taps.update(fc1=h1, logits=logits)
    act_np, pred_np = act.to(torch.float32).numpy(), pred.to(torch.int32).numpy()
    if not return_taps:
        return act_np, pred_np
    out = {}
    for k, v in taps.items():
        v = v.permute(0, 2, 1) if (k.startswith("stem") or k.startswith("module")) else v
        out[k] = v.contiguous().to(torch.float32).numpy()
    return act_np, pred_np, out
