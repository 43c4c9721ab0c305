"""Native feature-TSV reader and result-row formatter (scope row f1) — ctypes over
libdeepsignal_hip.so's ds_tsv_* / ds_format_rows entry points.

`FeatureReader` yields the same queue items as the reference reader
(/root/reference/deepsignal/call_modifications.py:35-91) but as numpy arrays, parsed by host threads;
`format_rows` writes the rows `_call_mods` builds one Python string at a time (:183-190)."""
from __future__ import annotations

import ctypes
from typing import Iterator, NamedTuple

import numpy as np

from .engine import load_library


def usable_cpus() -> int:
    """Host CPUs this process may use: affinity mask and cgroup-v2 quota both count (the GPU boxes expose 256 logical
    CPUs under a 16-CPU quota; a thread team wider than the quota only thrashes)."""
    import os
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


class FeatureItem(NamedTuple):
    """One queue item (all rows of f5_batch_num reads)."""
    info: np.ndarray          # uint8 buffer: the first six columns of every row, concatenated
    info_off: np.ndarray      # int64[n+1]
    kmer: np.ndarray          # int32[n, kmer_len]
    means: np.ndarray         # float32[n, kmer_len]
    stds: np.ndarray
    lens: np.ndarray          # float32 (event lengths, cast as the TF feed does)
    signals: np.ndarray       # float32[n, signal_len]
    labels: np.ndarray        # int32[n]

    def sampleinfo(self):
        b = self.info.tobytes()
        return [b[self.info_off[i]:self.info_off[i + 1]].decode() for i in range(len(self.labels))]

    def as_features_batch(self):
        """The reference's 7-tuple layout (lists) for code written against it."""
        return (self.sampleinfo(), self.kmer.tolist(), self.means.tolist(), self.stds.tolist(),
                self.lens.astype(np.int64).tolist(), self.signals.tolist(), self.labels.tolist())


def _bind():
    lib = load_library()
    if getattr(lib, "_io_bound", False):
        return lib
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    lib.ds_tsv_open.argtypes = [ctypes.c_char_p, i32, i32, i32, ctypes.POINTER(vp)]
    lib.ds_tsv_close.argtypes = [vp]
    lib.ds_tsv_close.restype = None
    lib.ds_tsv_error.argtypes = [vp]
    lib.ds_tsv_error.restype = ctypes.c_char_p
    lib.ds_tsv_next.argtypes = [vp, i32]
    lib.ds_tsv_next.restype = i64
    lib.ds_tsv_locate.argtypes = [vp, i32]
    lib.ds_tsv_locate.restype = i64
    lib.ds_tsv_parse_into.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp]
    lib.ds_tsv_parse_into.restype = i64
    for name in ("kmer", "means", "stds", "lens", "signals", "labels", "info", "info_offsets"):
        f = getattr(lib, "ds_tsv_" + name)
        f.argtypes = [vp]
        f.restype = vp
    lib.ds_tsv_size.argtypes = [vp]
    lib.ds_tsv_size.restype = i64
    lib.ds_tsv_align.argtypes = [vp, i64]
    lib.ds_tsv_align.restype = i64
    lib.ds_tsv_set_range.argtypes = [vp, i64, i64]
    lib.ds_format_rows.argtypes = [i64, vp, vp, vp, i32, vp, vp, i32, vp, i64]
    lib.ds_format_rows.restype = i64
    lib._io_bound = True
    return lib


def _view(ptr, dtype, shape):
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape, dtype)
    buf = (ctypes.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape).copy()


class FeatureReader:
    def __init__(self, path: str, kmer_len: int = 17, signal_len: int = 360, nthreads: int = 0):
        self._lib = _bind()
        self._h = ctypes.c_void_p()
        rc = self._lib.ds_tsv_open(path.encode(), kmer_len, signal_len, nthreads, ctypes.byref(self._h))
        if rc != 0:
            raise IOError("cannot open feature file %s (%d)" % (path, rc))
        self.kmer_len, self.signal_len = kmer_len, signal_len

    def close(self):
        if self._h.value:
            self._lib.ds_tsv_close(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def size(self) -> int:
        return int(self._lib.ds_tsv_size(self._h))

    def align(self, pos: int) -> int:
        """First read boundary at or after byte `pos` (ds_tsv_align): cut points every rank computes alike."""
        got = int(self._lib.ds_tsv_align(self._h, int(pos)))
        if got < 0:
            raise ValueError("ds_tsv_align(%d) failed (%d)" % (pos, got))
        return got

    def set_range(self, begin: int, end: int) -> None:
        """Read rows of the byte range [begin, end) only (both ends from align())."""
        if self._lib.ds_tsv_set_range(self._h, int(begin), int(end)) != 0:
            raise ValueError("bad byte range [%d, %d)" % (begin, end))

    def cut_points(self, nchunks: int) -> list:
        """nchunks + 1 byte offsets that tile the file into chunks of whole reads, near-equal in bytes."""
        size = self.size
        pts = [self.align(size * k // nchunks) for k in range(nchunks)] + [size]
        for k in range(1, len(pts)):                     # monotone by construction; keep it so explicitly
            pts[k] = max(pts[k], pts[k - 1])
        return pts

    def items(self, f5_batch_num: int = 50) -> Iterator[FeatureItem]:
        lib, h, K, S = self._lib, self._h, self.kmer_len, self.signal_len
        while True:
            # locate the item's rows, then parse them straight into arrays this item owns (no copy out of the reader)
            n = lib.ds_tsv_locate(h, f5_batch_num)
            if n < 0:
                raise ValueError("feature file: %s" % lib.ds_tsv_error(h).decode())
            if n == 0:
                return
            kmer, labels = np.empty((n, K), np.int32), np.empty((n,), np.int32)
            means, stds, lens = (np.empty((n, K), np.float32) for _ in range(3))
            signals = np.empty((n, S), np.float32)
            got = lib.ds_tsv_parse_into(h, n, kmer.ctypes.data, means.ctypes.data, stds.ctypes.data, lens.ctypes.data,
                                        signals.ctypes.data, labels.ctypes.data)
            if got != n:
                raise ValueError("feature file: %s" % lib.ds_tsv_error(h).decode())
            off = _view(lib.ds_tsv_info_offsets(h), np.int64, (n + 1,))
            yield FeatureItem(_view(lib.ds_tsv_info(h), np.uint8, (int(off[-1]),)), off, kmer, means, stds, lens, signals, labels)


def format_rows(item_info: np.ndarray, info_off: np.ndarray, act: np.ndarray, pred: np.ndarray,
                kmer: np.ndarray) -> bytes:
    """Rows of one batch as bytes (each ends with a newline)."""
    lib = _bind()
    n = int(pred.shape[0])
    if n == 0:
        return b""
    info = np.ascontiguousarray(item_info, np.uint8)
    off = np.ascontiguousarray(info_off, np.int64)
    act = np.ascontiguousarray(act, np.float32)
    pred = np.ascontiguousarray(pred, np.int32)
    kmer = np.ascontiguousarray(kmer, np.int32)
    cap = int(off[n] - off[0]) + n * (2 * 20 + 16 + kmer.shape[1] + 8) + 64
    out = np.empty(cap, np.uint8)
    base = off - off[0]
    got = lib.ds_format_rows(n, info.ctypes.data + int(off[0]), base.ctypes.data, act.ctypes.data, act.shape[1],
                             pred.ctypes.data, kmer.ctypes.data, kmer.shape[1], out.ctypes.data, cap)
    if got < 0:
        raise RuntimeError("ds_format_rows failed (%d)" % got)
    return out[:got].tobytes()
