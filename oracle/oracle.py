"""ctypes wrapper around oracle/ds_oracle.c — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(as the checker / the timed CPU baseline). The product package `deepsignal_amd` never does.

Parity unpinned by the reference (TensorFlow 1.x absent, no golden vectors: SURVEY.md F5/F7);
see ds_oracle.c for the per-function citations.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
NMOD = 11
NLAYER = 3


class _Taps(ctypes.Structure):
    _fields_ = [
        ("stem_pool", ctypes.c_void_p),
        ("stem_conv2", ctypes.c_void_p),
        ("stem_conv3", ctypes.c_void_p),
        ("module_out", ctypes.c_void_p * NMOD),
        ("signal_feat", ctypes.c_void_p),
        ("lstm_h", (ctypes.c_void_p * NLAYER) * 2),
        ("joint", ctypes.c_void_p),
        ("fc1", ctypes.c_void_p),
        ("logits", ctypes.c_void_p),
    ]


def build(force: bool = False) -> None:
    src = os.path.join(_HERE, "ds_oracle.c")
    libs = [os.path.join(_BUILD, "libds_oracle_f32.so"), os.path.join(_BUILD, "libds_oracle_f64.so")]
    stale = force or any((not os.path.exists(l)) or os.path.getmtime(l) < os.path.getmtime(src) for l in libs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "all"], stdout=subprocess.DEVNULL)


_LIBS: Dict[str, ctypes.CDLL] = {}


def _lib(precision: str) -> ctypes.CDLL:
    if precision not in ("f32", "f64"):
        raise ValueError(precision)
    if precision not in _LIBS:
        build()
        lib = ctypes.CDLL(os.path.join(_BUILD, "libds_oracle_%s.so" % precision))
        lib.ds_oracle_forward.restype = ctypes.c_int
        lib.ds_oracle_num_tensors.restype = ctypes.c_int
        _LIBS[precision] = lib
    return _LIBS[precision]


def usable_cores() -> int:
    """Cores this process may use (affinity mask and cgroup CPU quota respected): an OpenMP team wider than the
    quota thrashes on the GPU boxes (256 logical CPUs, 16-CPU quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def forward(weights: Dict[str, np.ndarray], feats: Dict[str, np.ndarray], precision: str = "f32",
            taps: bool = False, nthreads: int = 0, kmer_len: int = 17, signal_len: int = 360,
            class_num: int = 2, is_cnn: bool = True, is_rnn: bool = True, is_base: bool = True):
    """Run the oracle. Returns (act [n,class_num] f32, pred [n] i32[, taps dict])."""
    from deepsignal_amd import spec   # spec only (names/shapes); never the engine
    lib = _lib(precision)
    table = spec.tensor_table(kmer_len, signal_len, class_num, is_cnn=is_cnn, is_rnn=is_rnn, is_base=is_base)
    assert lib.ds_oracle_num_tensors(int(is_cnn), int(is_rnn), int(is_base)) == len(table)
    arrs = [np.ascontiguousarray(weights[name], dtype=np.float32) for name, _ in table]
    for a, (name, shape) in zip(arrs, table):
        assert tuple(a.shape) == tuple(shape), name
    ptrs = (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    kmer = np.ascontiguousarray(feats["kmer"], dtype=np.int32)
    n = kmer.shape[0]
    means = np.ascontiguousarray(feats["means"], dtype=np.float32)
    stds = np.ascontiguousarray(feats["stds"], dtype=np.float32)
    sanums = np.ascontiguousarray(feats["sanums"], dtype=np.float32)
    signals = np.ascontiguousarray(feats["signals"], dtype=np.float32)
    assert kmer.shape == (n, kmer_len) and signals.shape == (n, signal_len)
    act = np.empty((n, class_num), np.float32)
    pred = np.empty((n,), np.int32)
    d = spec.net_dims(kmer_len, signal_len, class_num, is_cnn, is_rnn)
    tap_arrays: Optional[Dict[str, np.ndarray]] = None
    tp = None
    if taps:
        tap_arrays = {
            "stem_pool": np.empty((n, d.w_a, 64), np.float32),
            "stem_conv2": np.empty((n, d.w_a, 128), np.float32),
            "stem_conv3": np.empty((n, d.w_a, 256), np.float32),
            "signal_feat": np.empty((n, d.signal_feat), np.float32),
            "joint": np.empty((n, d.joint), np.float32),
            "fc1": np.empty((n, d.joint), np.float32),
            "logits": np.empty((n, class_num), np.float32),
        }
        st = _Taps()
        for k in ("stem_pool", "stem_conv2", "stem_conv3", "signal_feat", "joint", "fc1", "logits"):
            setattr(st, k, tap_arrays[k].ctypes.data)
        for m in range(NMOD):
            a = np.empty((n, d.module_width(m + 1), 240), np.float32)
            tap_arrays["module%d" % (m + 1)] = a
            st.module_out[m] = a.ctypes.data
        for di, dn in enumerate(("fw", "bw")):
            for l in range(NLAYER):
                a = np.empty((n, kmer_len, 256), np.float32)
                tap_arrays["lstm_%s_l%d" % (dn, l)] = a
                st.lstm_h[di][l] = a.ctypes.data
        tp = ctypes.byref(st)
    rc = lib.ds_oracle_forward(ctypes.c_int(kmer_len), ctypes.c_int(signal_len), ctypes.c_int(class_num),
                               ctypes.c_int(int(is_cnn)), ctypes.c_int(int(is_rnn)), ctypes.c_int(int(is_base)),
                               ptrs, ctypes.c_int(n),
                               ctypes.c_void_p(kmer.ctypes.data), ctypes.c_void_p(means.ctypes.data),
                               ctypes.c_void_p(stds.ctypes.data), ctypes.c_void_p(sanums.ctypes.data),
                               ctypes.c_void_p(signals.ctypes.data), ctypes.c_void_p(act.ctypes.data),
                               ctypes.c_void_p(pred.ctypes.data), tp, ctypes.c_int(nthreads or usable_cores()))
    if rc != 0:
        raise RuntimeError("ds_oracle_forward failed: %d" % rc)
    if taps:
        if not is_cnn:
            tap_arrays = {k: v for k, v in tap_arrays.items() if not (k.startswith("stem") or k.startswith("module") or k == "signal_feat")}
        if not is_rnn:
            tap_arrays = {k: v for k, v in tap_arrays.items() if not k.startswith("lstm")}
        return act, pred, tap_arrays
    return act, pred
