"""ctypes binding of libdeepsignal_hip.so — the MI355X engine behind `call_mods`.

`Engine` plays the role of the reference's `(Model, tf.Session)` pair
(/root/reference/deepsignal/call_modifications.py:203-212): construct, restore weights, then
`run(...)` == `tf_sess.run([model.activation_logits, model.prediction], feed_dict)`
(call_modifications.py:168-178).

There is NO CPU fallback: if the HIP library is missing or no GPU is visible, construction raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DS_HIP_LIBRARY: developer override (kernel A/B builds under tools/). The product path is the in-tree library; an active
# override is announced on stderr when the library is loaded and shows up in bench.py's JSON ("library_override"), so a
# variable left over from an A/B session cannot silently change what a product run or a benchmark measures.
DEFAULT_LIB_PATH = os.path.join(_HERE, "libdeepsignal_hip.so")
LIB_PATH = os.environ.get("DS_HIP_LIBRARY") or DEFAULT_LIB_PATH
LIBRARY_OVERRIDE = LIB_PATH if os.path.abspath(LIB_PATH) != os.path.abspath(DEFAULT_LIB_PATH) else None

# every symbol include/deepsignal_hip.h declares
EXPORTED_SYMBOLS = (
    "ds_create", "ds_destroy", "ds_last_error", "ds_version", "ds_load_weights", "ds_set_tensor",
    "ds_finalize_weights", "ds_forward", "ds_forward_device", "ds_sync", "ds_alloc_host", "ds_free_host",
    "ds_get_intermediate", "ds_set_profiling", "ds_num_stages", "ds_get_stage", "ds_reset_stage_times",
    "ds_set_graph", "ds_num_kernels", "ds_get_kernel_stat", "ds_submit", "ds_submit_parts", "ds_wait", "ds_num_slots",
    # scope row f1 (host I/O)
    "ds_tsv_open", "ds_tsv_close", "ds_tsv_error", "ds_tsv_next", "ds_tsv_kmer", "ds_tsv_means", "ds_tsv_stds",
    "ds_tsv_lens", "ds_tsv_signals", "ds_tsv_labels", "ds_tsv_info", "ds_tsv_info_offsets", "ds_format_rows",
    "ds_tsv_size", "ds_tsv_align", "ds_tsv_set_range", "ds_tsv_locate", "ds_tsv_parse_into",
    # scope row f3 (TF checkpoint import)
    "ds_crc32c",
)


class DsConfig(ctypes.Structure):
    _fields_ = [
        ("kmer_len", ctypes.c_int32), ("signal_len", ctypes.c_int32), ("class_num", ctypes.c_int32),
        ("is_cnn", ctypes.c_int32), ("is_rnn", ctypes.c_int32), ("is_base", ctypes.c_int32),
        ("device", ctypes.c_int32), ("precision", ctypes.c_int32), ("max_batch", ctypes.c_int32),
        ("reserved", ctypes.c_int32 * 7),
    ]


# ds_config.precision (include/deepsignal_hip.h): "bf16" = bf16 conv + FC operands with fp32 accumulation, fp32 BiLSTM;
# "bf16_all" = also bf16 h / weight operands in the LSTM matmuls (fp32 accumulate, gates, cell state)
PRECISIONS = {"fp32": 0, "bf16": 1, "bf16_all": 2, "bf16x3": 3}
TUNE_NO_FUSED, TUNE_SERIAL, TUNE_DEBUG_STAMPS, TUNE_NO_FOLD_FC, TUNE_NO_CHAIN, TUNE_SHARED_EVENT_STREAM, TUNE_SPLIT_DENSE_NARROW, TUNE_NO_LSTM_XPROJ, TUNE_LSTM_XPROJ_ALL = 1, 2, 4, 8, 16, 32, 64, 128, 256     # ds_config.reserved[2]
LSTM_TILINGS = {"auto": 0, "narrow": 1, "wide": 2, "lds1": 3, "lds2": 4, "wide8": 5}     # ds_config.reserved[3]

_lib: Optional[ctypes.CDLL] = None


def _share_hip_runtime_with_torch() -> None:
    """One HIP runtime per process. PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME
    libamdhip64.so.7) and ask for it by file name, so if this library pulls in /opt/rocm's copy first, a later
    `import torch` loads a SECOND runtime that cannot see the GPU ("No HIP GPUs are available"). Mapping
    torch's copy first (no torch import needed) lets both resolve to the same runtime in either import order.
    Processes without torch installed simply use the system ROCm runtime."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load_library() -> ctypes.CDLL:
    """Load the in-tree HIP library; raise (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C deepsignal_amd/csrc`). There is no CPU fallback." % LIB_PATH)
    _share_hip_runtime_with_torch()
    if LIBRARY_OVERRIDE:
        import sys
        print("deepsignal_amd: DS_HIP_LIBRARY is set -- loading %s instead of the in-tree library" % LIBRARY_OVERRIDE, file=sys.stderr)
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    lib.ds_create.argtypes = [ctypes.POINTER(DsConfig), ctypes.POINTER(vp)]
    lib.ds_create.restype = ctypes.c_int
    lib.ds_destroy.argtypes = [vp]
    lib.ds_destroy.restype = None
    lib.ds_last_error.argtypes = [vp]
    lib.ds_last_error.restype = ctypes.c_char_p
    lib.ds_version.restype = ctypes.c_char_p
    lib.ds_load_weights.argtypes = [vp, ctypes.c_char_p]
    lib.ds_set_tensor.argtypes = [vp, ctypes.c_char_p, vp, ctypes.POINTER(i64), i32]
    lib.ds_finalize_weights.argtypes = [vp]
    lib.ds_forward.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.ds_forward_device.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.ds_sync.argtypes = [vp]
    lib.ds_submit.argtypes = [vp, i32, vp, vp, vp, vp, vp, ctypes.POINTER(i32)]
    lib.ds_submit_parts.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, ctypes.POINTER(i32)]
    lib.ds_wait.argtypes = [vp, i32, vp, vp]
    lib.ds_alloc_host.argtypes = [ctypes.c_size_t, ctypes.POINTER(vp)]
    lib.ds_free_host.argtypes = [vp]
    lib.ds_get_intermediate.argtypes = [vp, ctypes.c_char_p, vp, i64]
    lib.ds_get_intermediate.restype = i64
    lib.ds_set_profiling.argtypes = [vp, i32]
    lib.ds_num_stages.argtypes = [vp]
    lib.ds_get_stage.argtypes = [vp, i32, ctypes.c_char_p, i32, ctypes.POINTER(i32), ctypes.POINTER(ctypes.c_double),
                                 ctypes.POINTER(i64), ctypes.POINTER(ctypes.c_double)]
    lib.ds_reset_stage_times.argtypes = [vp]
    lib.ds_set_graph.argtypes = [vp, i32]
    lib.ds_num_kernels.argtypes = [vp]
    lib.ds_get_kernel_stat.argtypes = [vp, i32, ctypes.c_char_p, i32, ctypes.POINTER(i64),
                                       ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    _lib = lib
    return lib


class Engine:
    """One handle == one GPU == one full weight replica."""

    def __init__(self, kmer_len: int = 17, signal_len: int = 360, class_num: int = 2, device: int = 0,
                 max_batch: int = 512, is_cnn: bool = True, is_rnn: bool = True, is_base: bool = True,
                 debug: bool = False, slots: int = 0, precision: str = "fp32", serial: bool = False,
                 no_fused: bool = False, debug_stamps: bool = False, fold_fc: bool = True, lstm_tiling: str = "auto",
                 fuse_max_spt: int = 0, fuse_min_tiles: int = 0, chain_modules: bool = True, shared_event_stream: bool = False,
                 split_dense_min_n: int = 0, split_dense_narrow: bool = False, lstm_xproj=True):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        if precision not in PRECISIONS:
            raise ValueError("precision must be one of %s" % (sorted(PRECISIONS),))
        self.precision = precision
        cfg = DsConfig(kmer_len, signal_len, class_num, int(is_cnn), int(is_rnn), int(is_base), device,
                       PRECISIONS[precision], max_batch)
        cfg.reserved[0] = 1 if debug else 0
        cfg.reserved[1] = slots          # forwards in flight for run_device (0 = engine default)
        # per-handle diagnostics / tuning (include/deepsignal_hip.h: DS_TUNE_*, DS_LSTM_TILING_*); the library reads
        # no environment variables
        cfg.reserved[2] = (TUNE_NO_FUSED if no_fused else 0) | (TUNE_SERIAL if serial else 0) | \
                          (TUNE_DEBUG_STAMPS if debug_stamps else 0) | (0 if fold_fc else TUNE_NO_FOLD_FC) | \
                          (0 if chain_modules else TUNE_NO_CHAIN) | (TUNE_SHARED_EVENT_STREAM if shared_event_stream else 0) | \
                          (TUNE_SPLIT_DENSE_NARROW if split_dense_narrow else 0) | \
                          (0 if lstm_xproj else TUNE_NO_LSTM_XPROJ) | (TUNE_LSTM_XPROJ_ALL if lstm_xproj == "all" else 0)
        if lstm_tiling not in LSTM_TILINGS:
            raise ValueError("lstm_tiling must be one of %s" % (sorted(LSTM_TILINGS),))
        cfg.reserved[3] = LSTM_TILINGS[lstm_tiling]
        cfg.reserved[4] = fuse_max_spt
        cfg.reserved[5] = fuse_min_tiles
        cfg.reserved[6] = split_dense_min_n
        rc = self._lib.ds_create(ctypes.byref(cfg), ctypes.byref(self._h))
        if rc != 0:
            msg = self._lib.ds_last_error(None).decode()
            self._h = ctypes.c_void_p()
            raise RuntimeError("ds_create failed (%d): %s" % (rc, msg))
        self.kmer_len, self.signal_len, self.class_num = kmer_len, signal_len, class_num
        self._lib.ds_num_slots.argtypes = [ctypes.c_void_p]
        self._slots = int(self._lib.ds_num_slots(self._h))
        self.device, self.max_batch = device, max_batch

    # -- lifecycle -----------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            self._lib.ds_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str) -> None:
        if rc < 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self._lib.ds_last_error(self._h).decode()))

    # -- weights (Saver.restore, call_modifications.py:210-211) --------------------------------
    def load_weights_file(self, path: str) -> None:
        self._check(self._lib.ds_load_weights(self._h, path.encode()), "ds_load_weights")

    def load_weights(self, weights: Dict[str, np.ndarray]) -> None:
        for name, arr in weights.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (ctypes.c_int64 * a.ndim)(*a.shape)
            self._check(self._lib.ds_set_tensor(self._h, name.encode(), a.ctypes.data, shape, a.ndim), "ds_set_tensor")
        self._check(self._lib.ds_finalize_weights(self._h), "ds_finalize_weights")

    # -- forward (sess.run, call_modifications.py:177-178) -------------------------------------
    def run(self, kmer, means, stds, sanums, signals) -> Tuple[np.ndarray, np.ndarray]:
        """Host arrays in, (activation_logits float32[n,class_num], prediction int32[n]) out."""
        kmer = np.ascontiguousarray(kmer, dtype=np.int32)
        n = kmer.shape[0] if kmer.ndim == 2 else 0
        means = np.ascontiguousarray(means, dtype=np.float32)
        stds = np.ascontiguousarray(stds, dtype=np.float32)
        sanums = np.ascontiguousarray(sanums, dtype=np.float32)     # int -> float cast as the TF feed does
        signals = np.ascontiguousarray(signals, dtype=np.float32)
        act = np.empty((n, self.class_num), np.float32)
        pred = np.empty((n,), np.int32)
        if n == 0:
            return act, pred
        if kmer.shape != (n, self.kmer_len) or means.shape != kmer.shape or stds.shape != kmer.shape \
                or sanums.shape != kmer.shape or signals.shape != (n, self.signal_len):
            raise ValueError("feature arrays have inconsistent shapes")
        rc = self._lib.ds_forward(self._h, n, kmer.ctypes.data, means.ctypes.data, stds.ctypes.data,
                                  sanums.ctypes.data, signals.ctypes.data, act.ctypes.data, pred.ctypes.data)
        self._check(rc, "ds_forward")
        return act, pred

    def run_device(self, n: int, d_kmer: int, d_means: int, d_stds: int, d_sanums: int, d_signals: int,
                   d_act: int, d_pred: int) -> None:
        """Raw device pointers (e.g. torch tensors' data_ptr()); asynchronous, call sync()."""
        rc = self._lib.ds_forward_device(self._h, n, d_kmer, d_means, d_stds, d_sanums, d_signals, d_act, d_pred)
        self._check(rc, "ds_forward_device")

    def sync(self) -> None:
        self._check(self._lib.ds_sync(self._h), "ds_sync")

    def submit(self, kmer, means, stds, sanums, signals) -> Tuple[int, int]:
        """Asynchronous run() of one batch (n <= max_batch): returns a ticket for wait(). Up to `slots` batches may be
        in flight; wait() them in submission order."""
        kmer = np.ascontiguousarray(kmer, dtype=np.int32)
        n = int(kmer.shape[0])
        arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in (means, stds, sanums, signals)]
        t = ctypes.c_int32()
        self._check(self._lib.ds_submit(self._h, n, kmer.ctypes.data, *(a.ctypes.data for a in arrs), ctypes.byref(t)),
                    "ds_submit")
        return (int(t.value), n)

    def submit_parts(self, parts) -> Tuple[int, int]:
        """submit() of one batch given as row segments: `parts` is a sequence of (kmer, means, stds, sanums, signals)
        array tuples; the rows are gathered into the pinned staging buffer by the library (no concatenated copy)."""
        k = len(parts)
        keep = [(np.ascontiguousarray(p[0], dtype=np.int32),) + tuple(np.ascontiguousarray(a, dtype=np.float32) for a in p[1:5])
                for p in parts]
        counts = (ctypes.c_int32 * k)(*[int(p[0].shape[0]) for p in keep])
        ptrs = [(ctypes.c_void_p * k)(*[p[j].ctypes.data for p in keep]) for j in range(5)]
        n = int(sum(counts))
        t = ctypes.c_int32()
        self._check(self._lib.ds_submit_parts(self._h, k, counts, *ptrs, ctypes.byref(t)), "ds_submit_parts")
        return (int(t.value), n)

    def wait(self, ticket: Tuple[int, int]) -> Tuple[np.ndarray, np.ndarray]:
        slot, n = ticket
        act = np.empty((n, self.class_num), np.float32)
        pred = np.empty((n,), np.int32)
        self._check(self._lib.ds_wait(self._h, slot, act.ctypes.data, pred.ctypes.data), "ds_wait")
        return act, pred

    @property
    def slots(self) -> int:
        return self._slots

    # -- diagnostics ---------------------------------------------------------------------------
    def intermediate(self, name: str, shape) -> np.ndarray:
        out = np.empty(shape, np.float32)
        got = self._lib.ds_get_intermediate(self._h, name.encode(), out.ctypes.data, out.size)
        self._check(int(got), "ds_get_intermediate(%s)" % name)
        if got != out.size:
            raise RuntimeError("intermediate %s: expected %d floats, got %d" % (name, out.size, got))
        return out

    def set_profiling(self, mode) -> None:
        """0/False off, 1 per-kernel runs, 2/True per launch (see include/deepsignal_hip.h)."""
        mode = 2 if mode is True else int(mode)
        self._check(self._lib.ds_set_profiling(self._h, mode), "ds_set_profiling")

    def set_graph(self, enable: bool) -> None:
        self._check(self._lib.ds_set_graph(self._h, int(enable)), "ds_set_graph")

    def reset_stage_times(self) -> None:
        self._check(self._lib.ds_reset_stage_times(self._h), "ds_reset_stage_times")

    def stage_times(self) -> List[dict]:
        out = []
        for i in range(self._lib.ds_num_stages(self._h)):
            name = ctypes.create_string_buffer(64)
            launches = ctypes.c_int32()
            ms = ctypes.c_double()
            calls = ctypes.c_int64()
            flops = ctypes.c_double()
            self._check(self._lib.ds_get_stage(self._h, i, name, 64, ctypes.byref(launches), ctypes.byref(ms),
                                               ctypes.byref(calls), ctypes.byref(flops)), "ds_get_stage")
            out.append({"name": name.value.decode(), "launches": launches.value, "total_ms": ms.value,
                        "calls": calls.value, "flops_per_site": flops.value})
        return out

    def kernel_stats(self) -> List[dict]:
        out = []
        for i in range(self._lib.ds_num_kernels(self._h)):
            name = ctypes.create_string_buffer(96)
            launches = ctypes.c_int64()
            ms = ctypes.c_double()
            flops = ctypes.c_double()
            self._check(self._lib.ds_get_kernel_stat(self._h, i, name, 96, ctypes.byref(launches), ctypes.byref(ms),
                                                     ctypes.byref(flops)), "ds_get_kernel_stat")
            out.append({"name": name.value.decode(), "launches": launches.value, "total_ms": ms.value,
                        "flops": flops.value})
        return out
