"""Child process of tests/test_sanitizers.py: runs with libasan preloaded and loads the ASan + UBSan builds of the host
C / C++ of the path (oracle/Makefile `asan`): the CPU oracle on a small forward with every tap, and the product's
feature-TSV reader / row formatter / CRC on well-formed input and on the malformed corpus in tests/golden/malformed_tsv.
Any sanitizer report aborts the process; the parent checks the exit code and the summary line."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

ASAN = os.path.join(ROOT, "oracle", "_build", "asan")


def run_oracle():
    from deepsignal_amd import synth, weights
    from oracle import oracle
    lib = ctypes.CDLL(os.path.join(ASAN, "libds_oracle_f32.so"))
    lib.ds_oracle_forward.restype = ctypes.c_int
    lib.ds_oracle_num_tensors.restype = ctypes.c_int
    oracle._LIBS["f32"] = lib                     # the wrapper's marshalling, the instrumented library underneath
    for kw in (dict(kmer_len=17, signal_len=360), dict(kmer_len=5, signal_len=40), dict(kmer_len=9, signal_len=101)):
        w = weights.random_weights(seed=2, lstm_bias_std=0.1, **kw)
        f = synth.synthetic_features(3, seed=1, **kw) if "kmer_len" in synth.synthetic_features.__code__.co_varnames else None
        if f is None:
            f = synth.synthetic_features(3, seed=1)
            f = {"kmer": f["kmer"][:, :kw["kmer_len"]].copy(), "means": f["means"][:, :kw["kmer_len"]].copy(),
                 "stds": f["stds"][:, :kw["kmer_len"]].copy(), "sanums": f["sanums"][:, :kw["kmer_len"]].copy(),
                 "signals": np.resize(f["signals"], (3, kw["signal_len"])).astype(np.float32), "labels": f["labels"]}
        act, pred, taps = oracle.forward(w, f, "f32", taps=True, nthreads=2, **kw)
        assert np.isfinite(act).all() and len(taps) >= 20
    for variant in (dict(is_cnn=False), dict(is_rnn=False), dict(is_base=False)):
        w = weights.random_weights(seed=3, **variant)
        f = synth.synthetic_features(2, seed=4)
        act, _ = oracle.forward(w, f, "f32", nthreads=1, **variant)
        assert np.isfinite(act).all()
    return "oracle ok"


def run_io():
    lib = ctypes.CDLL(os.path.join(ASAN, "libds_io.so"))
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    lib.ds_tsv_open.argtypes = [ctypes.c_char_p, i32, i32, i32, ctypes.POINTER(vp)]
    lib.ds_tsv_close.argtypes = [vp]; lib.ds_tsv_close.restype = None
    lib.ds_tsv_error.argtypes = [vp]; lib.ds_tsv_error.restype = ctypes.c_char_p
    lib.ds_tsv_next.argtypes = [vp, i32]; lib.ds_tsv_next.restype = i64
    lib.ds_tsv_size.argtypes = [vp]; lib.ds_tsv_size.restype = i64
    lib.ds_tsv_align.argtypes = [vp, i64]; lib.ds_tsv_align.restype = i64
    lib.ds_tsv_set_range.argtypes = [vp, i64, i64]
    for name in ("kmer", "means", "stds", "lens", "signals", "labels", "info", "info_offsets"):
        getattr(lib, "ds_tsv_" + name).argtypes = [vp]
        getattr(lib, "ds_tsv_" + name).restype = vp
    lib.ds_format_rows.argtypes = [i64, vp, vp, vp, i32, vp, vp, i32, vp, i64]
    lib.ds_format_rows.restype = i64
    lib.ds_crc32c.argtypes = [vp, ctypes.c_size_t, ctypes.c_uint32]
    lib.ds_crc32c.restype = ctypes.c_uint32
    d = os.path.join(ROOT, "tests", "golden", "malformed_tsv")
    man = json.load(open(os.path.join(d, "manifest.json")))
    K, S = man["kmer_len"], man["signal_len"]
    seen = {}
    for name, expect in sorted(man["files"].items()):
        for nthreads in (1, 3):
            h = vp()
            assert lib.ds_tsv_open(os.path.join(d, name).encode(), K, S, nthreads, ctypes.byref(h)) == 0
            size = lib.ds_tsv_size(h)
            cuts = [lib.ds_tsv_align(h, size * k // 7) for k in range(8)]        # boundary scan over garbage too
            assert all(0 <= c <= size for c in cuts)
            rows, status = 0, "ok"
            while True:
                n = lib.ds_tsv_next(h, 1)
                if n < 0:
                    status = "error"
                    assert len(lib.ds_tsv_error(h)) > 0
                    break
                if n == 0:
                    break
                rows += n
                # touch every output array the way the Python wrapper does, then format the rows
                off = np.frombuffer((ctypes.c_char * (8 * (n + 1))).from_address(lib.ds_tsv_info_offsets(h)), np.int64).copy()
                info = np.frombuffer((ctypes.c_char * int(off[-1])).from_address(lib.ds_tsv_info(h)), np.uint8).copy() if off[-1] else np.zeros(1, np.uint8)
                kmer = np.frombuffer((ctypes.c_char * (4 * n * K)).from_address(lib.ds_tsv_kmer(h)), np.int32).copy()
                sig = np.frombuffer((ctypes.c_char * (4 * n * S)).from_address(lib.ds_tsv_signals(h)), np.float32).copy()
                for acc in ("means", "stds", "lens"):
                    np.frombuffer((ctypes.c_char * (4 * n * K)).from_address(getattr(lib, "ds_tsv_" + acc)(h)), np.float32).sum()
                act = np.tile(np.array([[0.25, 0.75]], np.float32), (n, 1))
                act[0] = (np.float32(sig[0]) if np.isfinite(sig[0]) else 1e-30, 3e-8)
                pred = np.ones(n, np.int32)
                cap = int(off[-1]) + n * (2 * 20 + 16 + K + 8) + 64
                out = np.empty(cap, np.uint8)
                got = lib.ds_format_rows(n, info.ctypes.data, off.ctypes.data, act.ctypes.data, 2, pred.ctypes.data,
                                         kmer.ctypes.data, K, out.ctypes.data, cap)
                assert 0 < got <= cap
                assert lib.ds_format_rows(n, info.ctypes.data, off.ctypes.data, act.ctypes.data, 2, pred.ctypes.data,
                                          kmer.ctypes.data, K, out.ctypes.data, 8) < 0        # too small: -(needed)
            if status == "ok" and expect.startswith("ok"):
                # a second pass restricted to each aligned range must see the same number of rows
                tot = 0
                for a, b in zip(cuts[:-1], cuts[1:]):
                    assert lib.ds_tsv_set_range(h, a, max(a, b)) == 0
                    while True:
                        n = lib.ds_tsv_next(h, 2)
                        assert n >= 0
                        if n == 0:
                            break
                        tot += n
                assert tot == rows, (name, tot, rows)
            lib.ds_tsv_close(h)
            got = "%s:%d" % (status, rows) if status == "ok" else "error"
            if expect != "any":
                assert got == expect, (name, got, expect)
            seen[name] = got
    # float formatting corner cases through the formatter (denormals, inf, nan, exponents)
    vals = np.array([0.0, 1.0, 1e-5, 9.999999e-5, 1e-4, 123456.78, 1e15, 1e16, 3.4e38, 1e-45, np.inf, np.nan], np.float32)
    n = len(vals)
    act = np.stack([vals, np.ones(n, np.float32)], axis=1)
    off = np.arange(n + 1, dtype=np.int64) * 2
    info = np.frombuffer(b"ab" * n, np.uint8).copy()
    kmer = np.zeros((n, K), np.int32); kmer[0, 0] = 99; kmer[1, 1] = -5
    out = np.empty(4096, np.uint8)
    got = lib.ds_format_rows(n, info.ctypes.data, off.ctypes.data, act.ctypes.data, 2, np.zeros(n, np.int32).ctypes.data,
                             kmer.ctypes.data, K, out.ctypes.data, 4096)
    assert got > 0 and out[:got].tobytes().count(b"\n") == n
    buf = np.random.default_rng(0).integers(0, 256, 4099, dtype=np.uint8)
    assert lib.ds_crc32c(buf.ctypes.data + 1, 4097, 0) == lib.ds_crc32c(buf.ctypes.data + 1, 4097, 0)      # unaligned start
    return "io ok (%d corpus files)" % len(seen)


if __name__ == "__main__":
    print("SANITIZER-CHILD:", run_oracle(), "|", run_io())
