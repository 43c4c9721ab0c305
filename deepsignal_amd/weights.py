"""Weight container ("DSAMDW01") and deterministic random-init generator.

The reference restores a TensorFlow-V2 checkpoint (`call_modifications.py:210-211`); no trained
weights ship with it (README.md:85-95). This module provides

  * a flat, dependency-free weight file keyed by the TF variable names of SURVEY.md Appendix B.7
    (so a checkpoint importer maps 1:1), and
  * a TF-initializer-style random generator (glorot-uniform kernels, truncated-normal embedding,
    `model.py:61-62`) with *randomised* BN statistics so BN folding is actually exercised.

File layout (little endian):
  8 B   magic  b"DSAMDW01"
  u32   n_tensors
  per tensor: u16 name_len, name bytes (utf-8), u8 ndim, u32 dims[ndim], u64 offset, u64 nbytes
  ...   zero padding to a 64-byte boundary, then the fp32 payloads at their `offset`s (absolute)
"""
from __future__ import annotations

import struct
from typing import Dict, Iterable, List, Tuple

import numpy as np

from . import spec

MAGIC = b"DSAMDW01"

WEIGHT_SEED = 20190417


def _glorot_uniform(rng: np.random.Generator, shape: Tuple[int, ...]) -> np.ndarray:
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:  # HWIO conv kernel
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


def _truncated_normal(rng: np.random.Generator, shape, std: float) -> np.ndarray:
    x = rng.normal(0.0, std, size=shape)
    bad = np.abs(x) > 2 * std
    while bad.any():
        x[bad] = rng.normal(0.0, std, size=int(bad.sum()))
        bad = np.abs(x) > 2 * std
    return x.astype(np.float32)


def random_weights(seed: int = WEIGHT_SEED, kmer_len: int = 17, signal_len: int = 360,
                   class_num: int = 2, lstm_bias_std: float = 0.0,
                   randomize_bn: bool = True, is_cnn: bool = True, is_rnn: bool = True,
                   is_base: bool = True) -> Dict[str, np.ndarray]:
    """Random-init parameters in the canonical order of `spec.tensor_table`.

    lstm_bias_std=0 reproduces TF's zero LSTM bias; tests pass a non-zero value so the bias path
    is exercised."""
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = {}
    for name, shape in spec.tensor_table(kmer_len, signal_len, class_num, is_cnn=is_cnn, is_rnn=is_rnn, is_base=is_base):
        leaf = name.rsplit("/", 1)[-1]
        if name.endswith("embedding"):
            w = _truncated_normal(rng, shape, float(np.sqrt(2.0 / spec.VOCAB_SIZE)))
        elif leaf == "kernel":
            w = _glorot_uniform(rng, shape)
        elif leaf == "bias":
            w = (rng.normal(0.0, lstm_bias_std, size=shape) if lstm_bias_std > 0
                 else np.zeros(shape)).astype(np.float32)
        elif leaf == "gamma":
            w = (rng.uniform(0.5, 1.5, size=shape) if randomize_bn else np.ones(shape)).astype(np.float32)
        elif leaf == "beta":
            w = (rng.normal(0.0, 0.1, size=shape) if randomize_bn else np.zeros(shape)).astype(np.float32)
        elif leaf == "moving_mean":
            w = (rng.normal(0.0, 0.1, size=shape) if randomize_bn else np.zeros(shape)).astype(np.float32)
        elif leaf == "moving_variance":
            w = (rng.uniform(0.5, 1.5, size=shape) if randomize_bn else np.ones(shape)).astype(np.float32)
        else:
            raise AssertionError(name)
        out[name] = np.ascontiguousarray(w)
    return out


STRESS_SEED = 20260417


def stress_weights(seed: int = STRESS_SEED, lstm_scale: float = 3.5, lstm_bias_std: float = 0.6,
                   gamma_hi: float = 3.0, hot_frac: float = 0.1, head: "np.ndarray | None" = None,
                   **geometry) -> Dict[str, np.ndarray]:
    """Random weights at the scale a TRAINED model has, for parity tests outside the linear regime of `random_weights`
    (whose glorot-uniform LSTM kernels keep every gate pre-activation near 0 and every logit within +-0.7):

      * LSTM kernels x `lstm_scale`, LSTM bias ~ N(0, `lstm_bias_std`): gate pre-activations of std ~1.5, |h| up to 0.99
        (`layers.py:45-72`: the sigmoid / tanh of the cells leave their linear part);
      * BN gamma: a `hot_frac` share of the channels of every layer drawn from U(1.5, `gamma_hi`), the rest as
        `random_weights` (all channels at 3 would grow the eleven residual modules to 1e5, which no trained net does);
      * `head`: a `dense_1/kernel` [J, 2] to install (see `centred_head`), else the glorot one stays.

    Pure numpy, no forward pass: tests/golden/make_stress_golden.py computes the head once and commits it."""
    w = random_weights(seed=seed, lstm_bias_std=lstm_bias_std, **geometry)
    rng = np.random.default_rng(seed + 1)
    for name in w:
        leaf = name.rsplit("/", 1)[-1]
        if name.endswith("lstm_cell/kernel"):
            w[name] = (w[name] * np.float32(lstm_scale)).astype(np.float32)
        elif leaf == "gamma":
            g = w[name].copy()
            hot = rng.random(g.shape) < hot_frac
            g[hot] = rng.uniform(1.5, gamma_hi, size=int(hot.sum()))
            w[name] = g.astype(np.float32)
    if head is not None:
        install_head(w, head)
    return w


def centred_head(fc1: np.ndarray, direction: np.ndarray, logit_std: float, seed: int) -> np.ndarray:
    """A two-class `dense_1/kernel` whose labels are balanced on the batch that produced `fc1` [n, J] (the input of
    `layers.py:261-263`'s second dense layer, from ANY forward implementation): `direction` [J] has its component along
    the batch mean of fc1 removed (mean logit 0 -- the joint model has no bias to do that with), is scaled so that the
    logits have standard deviation `logit_std` over the batch, and the second column is its negative plus 2 % noise
    (anti-correlated columns: sigmoid outputs far apart, both labels present)."""
    fc1 = np.asarray(fc1, np.float64)
    col = np.asarray(direction, np.float64).copy()
    m = fc1.mean(axis=0)
    col -= m * (m @ col) / (m @ m)
    col *= logit_std / (fc1 @ col).std()
    noise = np.random.default_rng(seed).normal(0.0, 0.02 * np.abs(col).mean(), size=col.shape)
    return np.stack([col, -col + noise], axis=1).astype(np.float32)


def install_head(weights: Dict[str, np.ndarray], head: np.ndarray) -> None:
    head = np.ascontiguousarray(head, dtype=np.float32)
    if head.shape != weights["dense_1/kernel"].shape:
        raise ValueError("head has shape %s, dense_1/kernel %s" % (head.shape, weights["dense_1/kernel"].shape))
    weights["dense_1/kernel"] = head


def check_weights(weights: Dict[str, np.ndarray], kmer_len: int = 17, signal_len: int = 360,
                  class_num: int = 2, **variant) -> None:
    for name, shape in spec.tensor_table(kmer_len, signal_len, class_num, **variant):
        if name not in weights:
            raise KeyError("missing tensor %s" % name)
        if tuple(weights[name].shape) != tuple(shape):
            raise ValueError("tensor %s has shape %s, expected %s" % (name, weights[name].shape, shape))


def save_weights(path: str, weights: Dict[str, np.ndarray]) -> None:
    names = list(weights.keys())
    entries: List[bytes] = []
    header_len = len(MAGIC) + 4
    metas = []
    for n in names:
        a = weights[n]
        nb = n.encode("utf-8")
        header_len += 2 + len(nb) + 1 + 4 * a.ndim + 16
        metas.append((nb, a))
    off = (header_len + 63) // 64 * 64
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<I", len(names)))
        offsets = []
        for nb, a in metas:
            nbytes = int(a.size) * 4
            f.write(struct.pack("<H", len(nb)))
            f.write(nb)
            f.write(struct.pack("<B", a.ndim))
            f.write(struct.pack("<%dI" % a.ndim, *a.shape))
            f.write(struct.pack("<QQ", off, nbytes))
            offsets.append(off)
            off = (off + nbytes + 63) // 64 * 64
        for (nb, a), o in zip(metas, offsets):
            f.seek(o)
            f.write(np.ascontiguousarray(a, dtype="<f4").tobytes())


def load_weights(path: str) -> Dict[str, np.ndarray]:
    out: Dict[str, np.ndarray] = {}
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError("%s: not a DSAMDW01 weight file" % path)
        (n,) = struct.unpack("<I", f.read(4))
        metas = []
        for _ in range(n):
            (ln,) = struct.unpack("<H", f.read(2))
            name = f.read(ln).decode("utf-8")
            (nd,) = struct.unpack("<B", f.read(1))
            dims = struct.unpack("<%dI" % nd, f.read(4 * nd))
            off, nbytes = struct.unpack("<QQ", f.read(16))
            metas.append((name, dims, off, nbytes))
        for name, dims, off, nbytes in metas:
            f.seek(off)
            out[name] = np.frombuffer(f.read(nbytes), dtype="<f4").reshape(dims).copy()
    return out


def ordered(weights: Dict[str, np.ndarray], kmer_len: int = 17, signal_len: int = 360,
            class_num: int = 2, **variant) -> Iterable[np.ndarray]:
    for name, _ in spec.tensor_table(kmer_len, signal_len, class_num, **variant):
        yield weights[name]
