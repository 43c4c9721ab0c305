"""Machine-readable specification of the deepsignal call_mods network.

This is the frozen statement of *what* the hot path computes; every
other component (oracle, torch cross-check, HIP engine, weight container) is driven by it.

Reference anchors (paths relative to /root/reference):
  * wiring / embedding / head ........ deepsignal/model.py:25-108
  * BiLSTM ............................ deepsignal/layers.py:20-72, 142-173
  * BN wrapper ........................ deepsignal/layers.py:80-84
  * inception module .................. deepsignal/layers.py:87-139
  * inception net (stem + 11 modules).. deepsignal/layers.py:176-239
  * joint FC head ..................... deepsignal/layers.py:242-264

Nothing in here executes arithmetic; it only names tensors and shapes.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

VOCAB_SIZE = 1024        # model.py:19
EMBEDDING_SIZE = 128     # model.py:20
HIDDEN = 256             # model.py:51 (hidden_num=256)
LSTM_LAYERS = 3          # model.py:50 (layer_num=3)
BN_EPS = 1e-3            # tf.contrib.layers.batch_norm default epsilon (layers.py:80-84)
FORGET_BIAS = 1.0        # tf.contrib.rnn.LSTMCell default forget_bias (layers.py:49-50)
INCEPTION_TIMES = 16     # layers.py:87 (times=16)
N_INCEPTION = 11         # layers.py:205-232
MODEL_PREFIX = "model"   # model.py:27


def same_pad(in_len: int, k: int, stride: int) -> Tuple[int, int, int]:
    """TF 'SAME' padding rule -> (out_len, pad_left, pad_right)."""
    out = -(-in_len // stride)
    total = max((out - 1) * stride + k - in_len, 0)
    left = total // 2
    return out, left, total - left


@dataclass(frozen=True)
class ConvBN:
    """One `conv2d(use_bias=False) -> batch_norm [-> relu]` pair (H=1, so a 1-D conv over W)."""
    scope: str          # TF variable scope holding '<conv_name>/kernel' and '<bn_name>/*'
    conv_name: str
    bn_name: str
    k: int
    cin: int
    cout: int
    stride: int
    relu: bool

    @property
    def kernel_name(self) -> str:
        return "%s/%s/kernel" % (self.scope, self.conv_name)

    def bn_tensor(self, which: str) -> str:
        return "%s/%s/%s" % (self.scope, self.bn_name, which)

    @property
    def kernel_shape(self) -> Tuple[int, int, int, int]:
        return (1, self.k, self.cin, self.cout)   # HWIO, H = 1


BN_PARTS = ("beta", "gamma", "moving_mean", "moving_variance")


def stem_convs(prefix: str = MODEL_PREFIX) -> List[ConvBN]:
    s = prefix + "signalm"
    return [
        ConvBN(s + "conv_layer1", "conv", "bn", 7, 1, 64, 2, True),     # layers.py:183-188
        ConvBN(s + "conv_layer2", "conv", "bn", 1, 64, 128, 1, True),   # layers.py:192-197
        ConvBN(s + "conv_layer3", "conv", "bn", 3, 128, 256, 1, True),  # layers.py:198-203
    ]


# order of the ten conv+BN pairs inside one inception module; keys are used by every backend
INCEPTION_KEYS = ("b1", "b2", "b3a", "b3b", "b4a", "b4b", "b5s", "b5a", "b5b", "b5c")


def inception_convs(n: int, cin: int, prefix: str = MODEL_PREFIX) -> Dict[str, ConvBN]:
    """Conv+BN pairs of inception module n (1-based). layers.py:87-139."""
    t = INCEPTION_TIMES
    s = prefix + "signalm"
    root = "%sincp_layer%d/%s%d" % (s, n, s, n)
    return {
        "b1":  ConvBN(root + "branch1_maxpooling", "conv1a_1x1", "bn", 1, cin, 3 * t, 1, True),
        "b2":  ConvBN(root + "branch2_1x1", "conv0b_1x1", "bn", 1, cin, 3 * t, 1, True),
        "b3a": ConvBN(root + "branch3_1x3", "conv0c_1x1", "bn1", 1, cin, 2 * t, 1, True),
        "b3b": ConvBN(root + "branch3_1x3", "conv1c_1x3", "bn2", 3, 2 * t, 3 * t, 1, True),
        "b4a": ConvBN(root + "branch4_1x5", "conv0d_1x1", "bn1", 1, cin, 2 * t, 1, True),
        "b4b": ConvBN(root + "branch4_1x5", "conv1d_1x5", "bn2", 5, 2 * t, 3 * t, 1, True),
        "b5s": ConvBN(root + "branch5_residual_1x3", "convstem_1x1", "bn0", 1, cin, 3 * t, 1, False),
        "b5a": ConvBN(root + "branch5_residual_1x3", "conv0e_1x1", "bn1", 1, cin, 2 * t, 1, True),
        "b5b": ConvBN(root + "branch5_residual_1x3", "conv1e_1x3", "bn2", 3, 2 * t, 4 * t, 1, True),
        "b5c": ConvBN(root + "branch5_residual_1x3", "conv2e_1x1", "bn3", 1, 4 * t, 3 * t, 1, False),
    }


INCEPTION_OUT = 5 * 3 * INCEPTION_TIMES    # 240 channels (layers.py:139)


@dataclass(frozen=True)
class NetDims:
    kmer_len: int
    signal_len: int
    class_num: int
    w_conv1: int      # after conv1 (stride 2)
    w_a: int          # modules 1-3   (after maxpool1)
    w_b: int          # modules 4-8   (after maxpool2)
    w_c: int          # modules 9-11  (after maxpool3)
    pad_conv1: Tuple[int, int]
    pad_pool1: Tuple[int, int]
    pad_pool2: Tuple[int, int]
    pad_pool3: Tuple[int, int]
    signal_feat: int  # w_c * 240
    event_feat: int   # 2 * HIDDEN
    joint: int

    def module_width(self, n: int) -> int:
        return self.w_a if n <= 3 else (self.w_b if n <= 8 else self.w_c)

    def module_cin(self, n: int) -> int:
        return 256 if n == 1 else INCEPTION_OUT


def net_dims(kmer_len: int = 17, signal_len: int = 360, class_num: int = 2,
             is_cnn: bool = True, is_rnn: bool = True) -> NetDims:
    """Shapes of the graph. is_cnn / is_rnn select which sub-model feeds the joint FC
    (model.py:28-29,89-95); at least one must be on."""
    if not (is_cnn or is_rnn):
        raise ValueError("at least one of is_cnn/is_rnn should be True")     # model.py:28-29
    w1, l1, r1 = same_pad(signal_len, 7, 2)
    wa, lp1, rp1 = same_pad(w1, 3, 2)
    wb, lp2, rp2 = same_pad(wa, 3, 2)
    wc, lp3, rp3 = same_pad(wb, 3, 2)
    sf = wc * INCEPTION_OUT
    ev = 2 * HIDDEN
    return NetDims(kmer_len, signal_len, class_num, w1, wa, wb, wc,
                   (l1, r1), (lp1, rp1), (lp2, rp2), (lp3, rp3), sf, ev,
                   (sf if is_cnn else 0) + (ev if is_rnn else 0))


def lstm_tensor(direction: str, layer: int, which: str, prefix: str = MODEL_PREFIX) -> str:
    """TF variable name of a BiLSTM tensor; direction in {'fw','bw'}, which in {'kernel','bias'}."""
    return "%sem/%s/multi_rnn_cell/cell_%d/lstm_cell/%s" % (prefix, direction, layer, which)


def lstm_input_size(layer: int, is_base: bool = True) -> int:
    if layer > 0:
        return HIDDEN
    return EMBEDDING_SIZE + 3 if is_base else 3             # model.py:63-75


def tensor_table(kmer_len: int = 17, signal_len: int = 360, class_num: int = 2,
                 prefix: str = MODEL_PREFIX, is_cnn: bool = True, is_rnn: bool = True,
                 is_base: bool = True) -> List[Tuple[str, Tuple[int, ...]]]:
    """Canonical ordered list of (TF variable name, shape) for the inference parameters.

    The order is the contract between weights.py, the oracle wrapper and the HIP engine.
    Variants (model.py:28-29,59-75,89-95): without is_rnn there is no embedding / LSTM; without
    is_base the LSTM reads only (mean, std, len); without is_cnn the signal model does not feed the
    joint FC (the reference still instantiates it, its tensors are simply not needed here)."""
    d = net_dims(kmer_len, signal_len, class_num, is_cnn, is_rnn)
    out: List[Tuple[str, Tuple[int, ...]]] = []
    if is_rnn:
        if is_base:
            out.append((prefix + "embedding", (VOCAB_SIZE, EMBEDDING_SIZE)))
        for direction in ("fw", "bw"):
            for layer in range(LSTM_LAYERS):
                out.append((lstm_tensor(direction, layer, "kernel", prefix),
                            (lstm_input_size(layer, is_base) + HIDDEN, 4 * HIDDEN)))
                out.append((lstm_tensor(direction, layer, "bias", prefix), (4 * HIDDEN,)))

    def add_conv(c: ConvBN) -> None:
        out.append((c.kernel_name, c.kernel_shape))
        for part in BN_PARTS:
            out.append((c.bn_tensor(part), (c.cout,)))

    if is_cnn:
        for c in stem_convs(prefix):
            add_conv(c)
        for n in range(1, N_INCEPTION + 1):
            convs = inception_convs(n, d.module_cin(n), prefix)
            for key in INCEPTION_KEYS:
                add_conv(convs[key])
    out.append(("dense/kernel", (d.joint, d.joint)))          # layers.py:257-259
    out.append(("dense_1/kernel", (d.joint, class_num)))      # layers.py:261-262
    return out


def param_count(**kw) -> int:
    n = 0
    for _, shape in tensor_table(**kw):
        c = 1
        for s in shape:
            c *= s
        n += c
    return n


# Algorithmic work per site
FLOPS_PER_SITE = 280_296_000
FLOPS_CONV_PER_SITE = 109_251_072
FLOPS_LSTM_PER_SITE = 98_250_752
FLOPS_FC_PER_SITE = 72_794_176
