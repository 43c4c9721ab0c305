#!/usr/bin/env python3
"""bench.py — CpG sites/sec of the call_mods forward on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (ds_forward_device) over one batch of 512 synthetic
(kmer=17, signal=360) sites whose features are already resident in HBM.  N>1: one process per GPU
(torch.distributed / RCCL), sites sharded by read with a full weight replica per rank (weak
scaling, no data-path collective); the only collective is the final result gather to rank 0.

`python bench.py --gpus N` with N > 1 and no launcher environment starts its own ranks: the parent spawns
`python -m torch.distributed.run --nproc-per-node N ... bench.py` as a CHILD process before it has imported torch or
touched the GPU, and exits with the child's code (a process that has initialised the GPU is never re-exec'ed).
`--dry-run` replaces the engine by a stub and RCCL by gloo: the launcher / barrier / gather / JSON plumbing runs on a
CPU-only box (tests/test_bench_launcher.py); its line carries "dry_run": true and is not a measurement.

The timed region is repeated `--windows` times (default 5), each window = EXACTLY K steps bracketed by barrier +
synchronize on both sides, max over ranks; `value` / `ms_per_step` are the MEDIAN window, `windows` lists them all.

Prints ONE JSON line on rank 0 (contract in the task statement). `value` / `ms_per_step` / `roofline` are the
REFERENCE GRAPH: the joint model runs as its three steps (avgpool kernel, dense 6032 x 6032, head; layers.py:233-238,
257-263), 280.3 MFLOP/site of which 244.0 are executed (layer 0's embedding product is a table lookup).
Extra objects:
  roofline            the dominant kernel's achieved TFLOP/s (algorithmic 2*M*N*K per launch / HIP-event duration per
                      launch) against the fp32 MFMA peak; fc_path = MFMA utilisation of the dense 6032 x 6032 GEMM,
                      conv_path = HBM GB/s of the signal model's kernels (the two figures the north star names)
  fast_mode_folded    the product's default engine (joint model folded into one 6032 x 2 matrix at weight load)
  cpu_baseline        the CPU oracle (oracle/ds_oracle.c, "port") timed on this box's host cores
  pcie_inclusive      the same 512-site batches from HOST memory through ds_submit / ds_wait (H2D + kernels + D2H,
                      SURVEY.md 8d's definition of the metric) -- reported beside `value`, never as it
  ds_forward_blocking the blocking call the reference makes (call_modifications.py:177-178; INTEGRATION.md binds it):
                      host buffers in, results out, n = 512 and n = 8192 per call
  e2e_tsv             feature TSV -> result TSV through call_mods (native reader + formatter + the default engine)
  configs2_bf16_batch4096   BASELINE configs[2]: bf16_all / bf16 throughput at batch 4096 and the conv path's HBM fraction
  e2e_tsv_sharded     N > 1 only: the product's multi-GPU route (by-read byte ranges of one feature TSV, per-rank engines, row text
                      gathered to rank 0 through OrderedRowGather), with the host-only parse ceiling of the same ranges beside it
  fp32_class_bf16x3   DS_PRECISION_BF16X3 (fp32 operands as three bf16 terms on the bf16 matrix pipe): the same steps, its own
                      roofline (bf16 MFMA peak / 6 products per MAC), accuracy against the float64 oracle on the stress set
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 512
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
CONV_BYTES_PER_SITE_MODULE_GRANULAR = 1_651_680      # conv path, fp32, module-granular
WORKLOAD = "configs[1]: 1xMI355X, batch=512, random-init CpG model weights, synthetic (17,360) features, fp32"


def effective_cores():
    """Host cores this process may actually use: the GPU boxes expose 256 logical CPUs but run the job under a cgroup
    CPU quota (cpu.max), and an OpenMP / MKL pool wider than the quota thrashes (measured: 256 threads are 2.6x
    slower than 32 for the C port, and PyTorch-CPU drops from 1,400 to 1.5 sites/s)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(-(-int(quota) // int(period)))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(weights, budget_s=12.0):
    """Bounded samples of the same workload through the two CPU stand-ins for the TF1-CPU path, both on
    all usable host cores: the C port (oracle/ds_oracle.c, OpenMP over sites) and the same graph in PyTorch-CPU
    library ops (oracle/torch_statement.py; oneDNN / MKL kernels, the closest analogue of TF1's Eigen / MKL ones).
    The reported baseline is the FASTER of the two; the other one rides along in `other_port`."""
    from deepsignal_amd import synth
    from oracle import oracle
    cores = effective_cores()
    nmax = 16384
    feats = synth.synthetic_features(nmax, seed=synth.FEATURE_SEED)
    nprobe = min(nmax, max(256, 8 * cores))
    probe = {k: v[:nprobe] for k, v in feats.items()}
    oracle.forward(weights, probe, "f32", nthreads=cores)          # warm-up (page-in, thread pool)
    t0 = time.perf_counter()
    oracle.forward(weights, probe, "f32", nthreads=cores)
    rate = nprobe / (time.perf_counter() - t0)
    n = int(min(nmax, max(nprobe, (rate * budget_s) // 512 * 512)))
    sample = {k: v[:n] for k, v in feats.items()}
    t0 = time.perf_counter()
    oracle.forward(weights, sample, "f32", nthreads=cores)
    dt = time.perf_counter() - t0
    c_port = {"value": round(n / dt, 2), "unit": "sites/s", "cores": cores, "kind": "port",
              "sample": "%d synthetic sites of the same workload (batches of 512) through oracle/ds_oracle.c "
                        "(f32, OpenMP, %d threads), %.1f s" % (n, cores, dt)}
    t_port = None
    try:
        import torch
        from oracle import torch_statement
        torch.set_num_threads(cores)
        wt = {k: torch.from_numpy(v) for k, v in weights.items()}
        t0 = time.perf_counter()
        torch_statement.forward(wt, {k: v[:64] for k, v in feats.items()}, dtype=torch.float32)    # warm-up + guard
        if time.perf_counter() - t0 > 20.0:
            raise RuntimeError("PyTorch-CPU probe of 64 sites took %.0f s; leg skipped" % (time.perf_counter() - t0))
        t0 = time.perf_counter()
        torch_statement.forward(wt, {k: v[:512] for k, v in feats.items()}, dtype=torch.float32)
        per = time.perf_counter() - t0
        reps = int(max(1, min(32, budget_s // max(per, 1e-3))))
        t0 = time.perf_counter()
        for r in range(reps):
            torch_statement.forward(wt, {k: v[512 * r:512 * (r + 1)] for k, v in feats.items()}, dtype=torch.float32)
        dt2 = time.perf_counter() - t0
        t_port = {"value": round(512 * reps / dt2, 2), "unit": "sites/s", "cores": cores, "kind": "port",
                  "sample": "%d synthetic sites of the same workload (batches of 512) through oracle/torch_statement.py "
                            "(PyTorch-CPU f32, %d threads), %.1f s" % (512 * reps, cores, dt2)}
    except Exception as exc:      # the baseline leg must not take the bench line down
        c_port["torch_cpu_error"] = repr(exc)
    if t_port is not None:
        best, other = (t_port, c_port) if t_port["value"] > c_port["value"] else (c_port, t_port)
        best["other_port"] = {"value": other["value"], "sample": other["sample"]}
        return best
    return c_port


def _latest_profile(pattern):
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return cands[-1] if cands else None


def _norm_kernel(s):
    return s.replace(" ", "").replace(",false>", ">").replace(",true>", ",bf16>")


def _build_of(rec):
    """Which build a committed PMC constant belongs to: the identity tools/build_identity.py wrote into the file when the counters
    were collected (commit of the snapshot + a hash of the kernel sources), next to the same hash of THIS run's sources."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from build_identity import sources_sha16
        now = sources_sha16()
    except Exception:
        now = None
    was = rec.get("sources_sha16")
    if was is None and rec.get("commit") is None:
        return "no build identity in the file (collected before round 5); this run's sources %s" % now
    return "collected at commit %s, sources %s; this run's sources %s%s" % (
        rec.get("commit"), was, now, "" if was is None or now is None else (" (same)" if was == now else " (DIFFERENT build)"))


def pmc_traffic(kernel_name, pattern="r[0-9][0-9]_pmc_traffic.json"):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are
    collected in separate runs of this same command and corrected as MI355X_MICROARCH.md prescribes; see
    tools/pmc_traffic.py). Counters cannot be read from inside the process: this is a COMMITTED CONSTANT of an earlier
    run, not a measurement of this one, and `traffic_source` says so."""
    path = _latest_profile(pattern)
    try:
        rec = json.load(open(path))
        for name, v in rec["kernels"].items():
            if _norm_kernel(name) == _norm_kernel(kernel_name):
                return {"traffic": round(v["hbm_bytes_per_launch"]), "traffic_unit": "B/launch",
                        "traffic_fetch": round(v["fetch_bytes_per_launch"]), "traffic_write": round(v["write_bytes_per_launch"]),
                        "traffic_source": "committed constant, NOT measured in this run: profiles/%s (%s; %s)"
                                          % (os.path.basename(path), _build_of(rec), rec["source"])}
    except (OSError, ValueError, KeyError, TypeError):
        pass
    return {"traffic": None}


def pmc_mfma_busy(kernel_name, pattern="r[0-9][0-9]_pmc_mfma_util.json"):
    """MFMA-pipe busy share of `kernel_name` (SQ_VALU_MFMA_BUSY_CYCLES / duration at the in-kernel clock) from the committed
    PMC pass -- a committed constant like pmc_traffic."""
    path = _latest_profile(pattern)
    try:
        rec = json.load(open(path))
        for name, v in rec["kernels"].items():
            if _norm_kernel(name) == _norm_kernel(kernel_name):
                return {"mfma_busy_pmc": v["mfma_busy_frac_at_inkernel_clock"],
                        # GRBM_GUI_ACTIVE / duration: the clock the chip ran THIS kernel at -- it lowers the clock under matrix load
                        # (2.40 GHz idle, 1.92 GHz with every CU on bf16 MFMAs: tools/attic/clock_probe.hip), and `peak` is quoted at 2.4
                        # (only for dispatches of >= 100 us and never above the part's 2.4 GHz: on 20 us launches the quotient over-counts)
                        "in_kernel_clock_ghz": (v.get("gui_active_over_duration_ghz")
                                                if (v.get("gui_active_over_duration_ghz") or 9.9) <= 2.4 and v.get("mean_us", 0) >= 100.0 else None),
                        "mfma_busy_definition": "SQ_VALU_MFMA_BUSY_CYCLES per SIMD / (dispatch duration x 2.17 GHz)",
                        "mfma_busy_source": "committed constant, NOT measured in this run: profiles/%s (%s)"
                                            % (os.path.basename(path), _build_of(rec))}
    except (OSError, ValueError, KeyError, TypeError):
        pass
    return {"mfma_busy_pmc": None}


class _DryRunEngine:
    """--dry-run only: no forward is executed; outputs are filled with a constant so the gather has something to move."""
    slots = 8

    def __init__(self, torch):
        self.torch = torch

    def run_device(self, n, dk, dm, ds_, dn, dg, d_act, d_pred):
        pass

    def sync(self):
        pass

    def close(self):
        pass


def self_launch(args):
    """--gpus N without a launcher environment: start N fresh ranks as a CHILD process tree. Nothing in this (parent)
    process has imported torch or made a HIP call at this point, and the parent never execs: it waits for the child
    and returns its exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: required for RCCL across processes on this pool
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def pcie_inclusive(eng, feats, steps, batch):
    """SURVEY.md 8d's metric boundary: the same batches from HOST memory through the asynchronous host boundary
    (ds_submit stages into pinned memory and enqueues H2D + forward + D2H; ds_wait hands the 12 B/site back), as many
    batches in flight as the engine has slots. Median of 3 windows of `steps` batches."""
    keys = ("kmer", "means", "stds", "sanums", "signals")
    npool = feats["kmer"].shape[0] // batch
    pool = [tuple(feats[k][b * batch:(b + 1) * batch] for k in keys) for b in range(npool)]
    rates = []
    for rep in range(4):                      # rep 0 = warm-up (pinned staging is allocated on first use)
        pending = []
        t0 = time.perf_counter()
        for i in range(steps):
            if len(pending) == eng.slots:
                eng.wait(pending.pop(0))
            pending.append(eng.submit(*pool[i % npool]))
        for t in pending:
            eng.wait(t)
        if rep:
            rates.append(steps * batch / (time.perf_counter() - t0))
    rates.sort()
    return {"value": round(rates[len(rates) // 2], 1), "unit": "sites/s", "min": round(rates[0], 1), "max": round(rates[-1], 1),
            "boundary": "ds_submit / ds_wait from host buffers: pinned staging + H2D (1,712 B/site) + forward + D2H "
                        "(12 B/site), %d batches of %d sites per window, %d in flight" % (steps, batch, eng.slots)}


def forward_blocking(eng, feats, batch):
    """The call the reference makes (tf_sess.run, call_modifications.py:177-178) and INTEGRATION.md binds: blocking
    ds_forward on host arrays, results back in host arrays. n = batch per call (ONE forward in flight: the latency of the
    19-launch BiLSTM chain is exposed) and n = 16 batches per call (the library pipelines the passes over its slots)."""
    keys = ("kmer", "means", "stds", "sanums", "signals")
    out = {"boundary": "ds_forward(host arrays) -> (act, pred) host arrays, blocking; pinned staging + one H2D / one D2H copy per pass"}
    for n in (batch, 16 * batch):
        reps_arrays = [tuple(np_tile(feats[k], n, off) for k in keys) for off in (0, batch)]
        eng.run(*reps_arrays[0])                                         # warm-up (plans, graphs, pinned staging)
        rates = []
        for rep in range(3):
            calls = max(2, (64 * batch) // n)
            t0 = time.perf_counter()
            for c in range(calls):
                eng.run(*reps_arrays[c & 1])
            rates.append(calls * n / (time.perf_counter() - t0))
        rates.sort()
        out["n_%d" % n] = {"value": round(rates[1], 1), "unit": "sites/s", "min": round(rates[0], 1), "max": round(rates[2], 1),
                           "ms_per_call": round(1e3 * n / rates[1], 4)}
    return out


def configs2_leg(Engine, w, torch, dev, local_rank, host_path=True):
    """BASELINE configs[2] inside the driver-run line: bf16 conv + FC with fp32 BiLSTM accumulate, batch 4096 (`bf16_all`: bf16
    operands everywhere, fp32 accumulate / gates / cell state; `bf16`: fp32 BiLSTM operands as well), resident inputs, 3
    windows of 30 steps, plus the stand-alone time of the fused inception kernel -> the conv path against the HBM roofline
    (module-granular bytes: 992 B per module row, 564 module rows per site)."""
    from deepsignal_amd import synth
    B, steps = 4096, 30
    keys = ("kmer", "means", "stds", "sanums", "signals")
    feats = synth.synthetic_features(B, seed=synth.FEATURE_SEED + 7)
    d = {k: torch.from_numpy(feats[k]).to(dev) for k in keys}
    act = torch.zeros((B, 2), dtype=torch.float32, device=dev)
    pred = torch.zeros((B,), dtype=torch.int32, device=dev)
    out = {"batch": B, "workload": "configs[2]: 1xMI355X, bf16 conv+FC with fp32 BiLSTM accumulate, batch=4096 (tolerance vs fp32: "
                                   "the tolerance_vs_fp32 object, gated by tests/test_gpu_bf16.py)"}
    for prec in ("bf16_all", "bf16"):
        e = Engine(device=local_rank, max_batch=B, precision=prec)
        e.load_weights(w)
        step = lambda: e.run_device(B, *(d[k].data_ptr() for k in keys), act.data_ptr(), pred.data_ptr())
        for _ in range(6):
            step()
        e.sync()
        rates = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            e.sync()
            rates.append(steps * B / (time.perf_counter() - t0))
        rates.sort()
        r = {"value": round(rates[1], 1), "unit": "sites/s", "ms_per_step": round(1e3 * B / rates[1], 4),
             "min": round(rates[0], 1), "max": round(rates[2], 1)}
        if prec == "bf16_all":
            e.set_profiling(3)                      # every launch on one stream: stand-alone kernel durations (HIP events)
            e.reset_stage_times()
            for _ in range(6):
                step()
            e.sync()
            ks = {k["name"]: k for k in e.kernel_stats() if k["launches"]}
            e.set_profiling(0)
            r["kernels_us_per_step_alone"] = {n: round(1e3 * k["total_ms"] / 6, 1) for n, k in ks.items()}
            conv = [k for n, k in ks.items() if n.startswith("inception_fused_bf16_kernel")]
            if conv:
                c_ms = sum(k["total_ms"] for k in conv) / 6
                c_bytes = 992 * 564 * B
                out["conv_path_hbm"] = {
                    "kernel": "inception_fused_bf16_kernel (3 launches: the modules of a width class chained per tile, the tile's rows "
                              "kept in LDS from module to module)",
                    "us_per_step": round(c_ms * 1e3, 1), "algorithmic_bytes": c_bytes, "achieved": round(c_bytes / (c_ms * 1e-3) / 1e9, 1),
                    "unit": "GB/s", "peak": 8000.0, "frac": round(c_bytes / (c_ms * 1e-3) / 8e12, 4),
                    "hbm_side_frac": None,      # counter-measured bytes / time / 8 TB/s (filled in below from the committed PMC pass)
                    "bf16_mfma_frac": round(sum(k["flops"] for k in conv) / 6 / (c_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                    "how": "module-granular bytes (SURVEY.md 8d): 992 B per module row (512 in + 480 out, bf16) x 564 module rows per "
                           "site x 4096 sites / HIP-event duration, every launch on one stream. The kernel moves fewer bytes than that: "
                           "only a chain's first module reads rows from HBM and only its last one writes them (traffic_per_step)"}
                t = pmc_traffic("inception_fused_bf16_kernel<3>", "r[0-9][0-9]_bf16_all_4096_pmc_traffic.json")
                if t.get("traffic"):
                    out["conv_path_hbm"].update({"traffic_per_step": 3 * t["traffic"], "traffic_source": t["traffic_source"],
                                                 "hbm_side_frac": round(3 * t["traffic"] / (c_ms * 1e-3) / 8e12, 4)})
        assert bool(torch.isfinite(act).all())
        if prec == "bf16_all" and host_path:
            # feature TSV -> result TSV through call_mods with this engine: the row pipeline fills the engine's 4096-site batches
            r["e2e_tsv"] = e2e_tsv(e, feats, 327680, B)
        out[prec] = r
        e.close()
    out["tolerance_vs_fp32"] = configs2_tolerance(Engine, w, local_rank, feats, keys)
    return out


def configs2_tolerance(Engine, w, local_rank, feats, keys):
    """configs[2]'s "tolerance vs fp32 reported", measured in this run on one 4096-site batch, in two columns: the benchmark
    weights (glorot-uniform: every logit within +-0.7, one label for the whole batch -- bf16 looks harmless there) and the
    trained-regime stress set of the parity tests (weights.stress_weights + the committed centred head: saturating LSTM
    gates, hot BN channels, logits spanning +-10, both labels; tests/test_gpu_bf16.py gates the same numbers)."""
    import numpy as np
    from deepsignal_amd import weights as W
    args = [feats[k] for k in keys]
    g = np.load(os.path.join(ROOT, "tests", "golden", "stress_golden.npz"))
    sets = (("benchmark_weights", w), ("stress_weights", W.stress_weights(int(g["stress_seed"]), head=g["stress_head"])))
    pn = lambda a: a / a.sum(axis=1, keepdims=True)
    out = {"how": "one 4096-site batch through ds_forward per precision and weight set; d = |sigmoid outputs (bf16 mode) - (fp32)|, "
                  "p_norm = act / (act0 + act1) as call_modifications.py:185-187 forms it; both sides with the folded joint model"}
    for name, ws in sets:
        e = Engine(device=local_rank, max_batch=len(args[0]), slots=1)
        e.load_weights(ws)
        a32, p32 = e.run(*args)
        e.close()
        ac = np.clip(a32.astype(np.float64), 1e-7, 1.0 - 1e-7)
        lg = np.log(ac) - np.log1p(-ac)
        col = {"label1_share_fp32": round(float(p32.mean()), 4), "logit_span_fp32": [round(float(lg.min()), 2), round(float(lg.max()), 2)]}
        for prec in ("bf16_all", "bf16"):
            e = Engine(device=local_rank, max_batch=len(args[0]), slots=1, precision=prec)
            e.load_weights(ws)
            a16, p16 = e.run(*args)
            e.close()
            d = np.abs(a16 - a32).max(axis=1)
            flips = p16 != p32
            margin = np.abs(a32[:, 1] - a32[:, 0])
            col[prec] = {"max_abs_d_act": float("%.3g" % d.max()), "mean_abs_d_act": float("%.3g" % d.mean()),
                         "max_abs_d_pnorm": float("%.3g" % np.abs(pn(a16) - pn(a32)).max()),
                         "label_flip_rate": round(float(flips.mean()), 5),
                         "largest_fp32_margin_of_a_flipped_site": float("%.3g" % (margin[flips].max() if flips.any() else 0.0))}
        out[name] = col
    return out


PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA" (dense)
SPLIT_PRODUCTS_PER_MAC = 6         # DS_PRECISION_BF16X3: three bf16 terms per operand, six term products per fp32-class MAC


def bf16x3_leg(Engine, w, torch, make_step, timed_windows, local_rank, slots, lstm_tiling, steps, host_feats=None):
    """DS_PRECISION_BF16X3 beside the headline: the same 512-site steps, the same timed-window rules, on an
    engine that carries fp32 operands as three bf16 terms through the bf16 matrix pipe (six products per MAC, fp32 accumulate) in
    the layers ds_version() lists, everything else as the fp32 engine. Its own key with its own roofline: algorithmic FLOPs of the
    split kernel per launch / stand-alone HIP-event duration against the bf16 MFMA peak / 6 products per MAC; the fp32-peak
    fraction rides along (above 1 = faster than a perfect native-fp32 kernel could be). Accuracy in the same run: one 512-site
    batch of the trained-regime stress set against the FLOAT64 CPU oracle, next to the native fp32 engine on the same batch."""
    import numpy as np
    from deepsignal_amd import synth, weights as W
    from oracle import oracle
    out = {"precision": "bf16x3", "what": "fp32 operands as three bf16 terms, six term products per MAC, fp32 accumulate "
                                          "(include/deepsignal_hip.h DS_PRECISION_BF16X3; deepsignal_amd/csrc/ds_split.hip)"}
    for key, fold in (("three_step", False), ("folded", True)):
        e = Engine(device=local_rank, max_batch=BATCH, lstm_tiling=lstm_tiling, slots=slots, fold_fc=fold, precision="bf16x3")
        e.load_weights(w)
        wins = timed_windows(e, 3)
        el = sorted(wins)[len(wins) // 2]
        out[key] = {"value": round(steps * BATCH / el, 1) if steps else 0.0, "unit": "sites/s", "ms_per_step": round(1e3 * el / max(steps, 1), 4)}
        if not fold:
            step = make_step(e)
            KP = min(steps, 50)
            e.set_profiling(3)
            e.reset_stage_times()
            for i in range(KP):
                step(i, i)
            e.sync()
            ks = {k["name"]: k for k in e.kernel_stats() if k["launches"]}
            e.set_profiling(0)
            out["kernels_us_per_step_alone"] = {n: round(1e3 * k["total_ms"] / max(KP, 1), 1) for n, k in ks.items()}
            sp = [k for n, k in ks.items() if "split" in n]
            if sp:
                k = max(sp, key=lambda k_: k_["total_ms"])
                per_launch = k["flops"] / k["launches"]
                avg_ms = k["total_ms"] / k["launches"]
                ach = per_launch / (avg_ms * 1e-3) / 1e12
                peak = PEAK_BF16_MFMA_TFLOPS / SPLIT_PRODUCTS_PER_MAC
                out["roofline"] = {"bound": "mfma", "kernel": k["name"], "launches_per_step": k["launches"] // max(KP, 1),
                                   "avg_launch_us": round(avg_ms * 1e3, 2), "flops_per_launch": per_launch,
                                   "achieved": round(ach, 2), "unit": "TFLOP/s (fp32-class MACs x 2)",
                                   "peak": round(peak, 1), "frac": round(ach / peak, 4),
                                   "peak_how": "bf16 dense MFMA peak %.0f TFLOP/s / %d term products per MAC" % (PEAK_BF16_MFMA_TFLOPS, SPLIT_PRODUCTS_PER_MAC),
                                   "frac_of_fp32_mfma_peak": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                                   "bf16_mfma_frac": round(ach * SPLIT_PRODUCTS_PER_MAC / PEAK_BF16_MFMA_TFLOPS, 4),
                                   "measured": "stand-alone: every launch of the step on ONE stream, HIP events on that stream, eager replay of %d steps" % KP}
                out["roofline"]["standalone_fracs"] = {
                    k_["name"]: round(k_["flops"] / (k_["total_ms"] * 1e-3) / 1e12 / peak, 4) for k_ in sp if k_["total_ms"] and k_["flops"]}
                out["roofline"].update(pmc_traffic(k["name"], "r[0-9][0-9]_bf16x3_pmc_traffic.json"))
                out["roofline"].update(pmc_mfma_busy(k["name"], "r[0-9][0-9]_bf16x3_pmc_mfma_util.json"))
        e.close()
    g = np.load(os.path.join(ROOT, "tests", "golden", "stress_golden.npz"))
    ws = W.stress_weights(int(g["stress_seed"]), head=g["stress_head"])
    f = synth.synthetic_features(BATCH, seed=931)
    args = [f[k] for k in ("kmer", "means", "stds", "sanums", "signals")]
    a64, p64 = oracle.forward(ws, f, "f64")
    acc = {"how": "one %d-site batch of the trained-regime stress set (tests/test_gpu_stress.py) through ds_forward per precision; "
                  "d = |sigmoid outputs - float64 CPU oracle|; both engines with the three-step joint model" % BATCH}
    for prec in ("fp32", "bf16x3"):
        e = Engine(device=local_rank, max_batch=BATCH, slots=1, fold_fc=False, precision=prec)
        e.load_weights(ws)
        a, p_ = e.run(*args)
        e.close()
        d = np.abs(a.astype(np.float64) - a64)
        acc[prec] = {"max_abs_d_act_vs_f64": float("%.3g" % d.max()), "mean_abs_d_act_vs_f64": float("%.3g" % d.mean()),
                     "label_flip_rate_vs_f64": round(float((p_ != p64).mean()), 5)}
    acc["label1_share_f64"] = round(float(p64.mean()), 4)
    out["accuracy_stress_set"] = acc
    if host_feats is not None:
        # feature TSV -> result TSV as `deepsignal call_mods --precision bf16x3` runs it (engine created for ENGINE_BATCH sites per forward)
        from deepsignal_amd import call_modifications as _cm
        e = Engine(device=local_rank, max_batch=_cm.ENGINE_BATCH["bf16x3"], precision="bf16x3")
        e.load_weights(w)
        out["e2e_tsv"] = e2e_tsv(e, host_feats, 163840, BATCH)
        e.close()
    return out


def np_tile(a, n, off):
    """n rows of `a` starting at row `off`, wrapping around (contiguous copy)."""
    import numpy as np
    idx = (np.arange(n) + off) % a.shape[0]
    return np.ascontiguousarray(a[idx])


def write_feature_tsv(path, feats, rows):
    """`rows` feature rows (12 columns, 20 sites per read) cycling over the synthetic sites of `feats`."""
    from deepsignal_amd.utils.process_utils import code2base_dna
    nuniq = min(rows, feats["kmer"].shape[0])
    tails = []
    for i in range(nuniq):          # columns 7..12 of a row; the first six are cheap and written per copy below
        tails.append("\t".join(["".join(code2base_dna[int(c)] for c in feats["kmer"][i]),
                                ",".join("%.6f" % x for x in feats["means"][i]), ",".join("%.6f" % x for x in feats["stds"][i]),
                                ",".join(str(int(x)) for x in feats["sanums"][i]),
                                ",".join("%.6f" % x for x in feats["signals"][i]), "1"]))
    with open(path, "w") as f:
        for i in range(rows):
            f.write("chr1\t%d\t+\t%d\tread_%06d\tt\t%s\n" % (1000 + i, i, i // 20, tails[i % nuniq]))
    return os.path.getsize(path)


def e2e_tsv_sharded(Engine, w, feats, dist, rank, world, gpu_index, cdev, rows_per_rank, batch):
    """The PRODUCT's multi-GPU route inside the scaling run: feature TSV -> result TSV through
    call_mods(dist=...) -- every rank parses only its own byte ranges of ONE shared file (cut at read boundaries), runs them on its
    own engine (created as `deepsignal call_mods` creates it), formats its own rows, and the row text travels to rank 0 through
    sharding.OrderedRowGather's collectives (reference topology: call_modifications.py:461-476, README.rst:15). Weak scaling:
    rows_per_rank x world rows. Also measured, so that a flat curve is attributable: the HOST-ONLY parse rate of the same byte
    ranges with the same per-rank thread count (the ceiling the 16-CPU cgroup quota of a box puts on TSV -> TSV)."""
    import contextlib
    import shutil
    import tempfile
    import torch
    from deepsignal_amd import call_modifications as cm, fastio
    rows = rows_per_rank * world
    box = [None]
    if rank == 0:
        tmpdir = tempfile.mkdtemp(prefix="ds_bench_sharded_")
        size = write_feature_tsv(os.path.join(tmpdir, "features.tsv"), feats, rows)
        box[0] = (tmpdir, size)
    dist.broadcast_object_list(box, src=0)
    tmpdir, size = box[0]
    path, out = os.path.join(tmpdir, "features.tsv"), os.path.join(tmpdir, "calls.tsv")
    threads = max(1, fastio.usable_cpus() // cm._local_world(world))
    try:
        # host-only: parse this rank's byte ranges (what _call_mods_sharded's reader does), nothing else
        rd = fastio.FeatureReader(path, 17, 360, nthreads=threads)
        per_rank = max(1, min(4096, rd.size // max(1, world * cm.SHARD_CHUNK_BYTES)))
        cuts = rd.cut_points(world * per_rank)
        t0 = time.perf_counter()
        n_parsed = 0
        for c in range(rank, world * per_rank, world):
            rd.set_range(cuts[c], cuts[c + 1])
            for item in rd.items(50):
                n_parsed += len(item.labels)
        t_parse = time.perf_counter() - t0
        rd.close()
        eng = Engine(device=gpu_index, max_batch=cm.ENGINE_BATCH["fp32"])
        eng.load_weights(w)
        times = []
        with contextlib.redirect_stdout(sys.stderr):
            for rep in range(2):               # rep 0 = warm-up (page cache, plans, pinned staging)
                dist.barrier()
                t0 = time.perf_counter()
                n = cm.call_mods(path, "unused", out, 17, 360, batch, 0.001, 2, 1, True, True, True, True, None, engine=eng, dist=dist,
                                 force_sharded=True)
                dist.barrier()
                times.append(time.perf_counter() - t0)
        eng.close()
        stat = torch.tensor([times[-1], t_parse, float(n_parsed)], dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(stat) for _ in range(world)]
        dist.all_gather(every, stat)
        every = [e.cpu().tolist() for e in every]
        ok = (n == rows) and (rank != 0 or sum(1 for _ in open(out)) == rows)
    finally:
        dist.barrier()
        if rank == 0:
            shutil.rmtree(tmpdir, ignore_errors=True)
    el = max(e[0] for e in every)
    parse_rates = [e[2] / e[1] if e[1] > 0 else 0.0 for e in every]
    return {"value": round(rows / el, 1), "unit": "sites/s", "rows": rows, "rows_per_rank": rows_per_rank, "input_MB": round(size / 1e6, 1),
            "seconds": round(el, 3), "complete": bool(ok), "engine": "default (folded joint model), fp32, max_batch %d" % cm.ENGINE_BATCH["fp32"],
            "parse_threads_per_rank": threads, "usable_cpus": fastio.usable_cpus(),
            "host_parse_only": {"sites_per_s_per_rank": [round(r, 1) for r in parse_rates], "sites_per_s_all_ranks": round(sum(parse_rates), 1),
                                "us_per_row_per_thread": round(1e6 * threads / max(parse_rates[0], 1e-9), 3),
                                "what": "the same byte ranges through the native reader alone, same thread count: the ceiling of TSV -> TSV on "
                                        "this box's host cores (all ranks parsing at once would share the CPU quota: the sum is an upper bound)"},
            "path": "call_mods(dist=process group): by-read byte ranges -> native reader -> ds_submit / ds_wait -> native formatter -> "
                    "OrderedRowGather (row text to rank 0 over %s) -> rank 0 writes in file order" % dist.get_backend()}


def e2e_tsv(eng, feats, rows, batch):
    """Feature TSV -> result TSV through the product's call_mods (native reader on the usable host cores, engine through
    submit / wait, native row formatter): what a user's `deepsignal call_mods -i features.tsv` sustains on this box."""
    import tempfile
    from deepsignal_amd import call_modifications as cm
    from deepsignal_amd.utils.process_utils import code2base_dna
    nuniq = min(rows, feats["kmer"].shape[0])
    tmpdir = tempfile.mkdtemp(prefix="ds_bench_")
    path = os.path.join(tmpdir, "features.tsv")
    tails = []
    for i in range(nuniq):          # columns 7..12 of a row; the first six are cheap and written per copy below
        tails.append("\t".join(["".join(code2base_dna[int(c)] for c in feats["kmer"][i]),
                                ",".join("%.6f" % x for x in feats["means"][i]), ",".join("%.6f" % x for x in feats["stds"][i]),
                                ",".join(str(int(x)) for x in feats["sanums"][i]),
                                ",".join("%.6f" % x for x in feats["signals"][i]), "1"]))
    with open(path, "w") as f:
        for i in range(rows):
            f.write("chr1\t%d\t+\t%d\tread_%06d\tt\t%s\n" % (1000 + i, i, i // 20, tails[i % nuniq]))
    import contextlib
    try:
        out = os.path.join(tmpdir, "calls.tsv")
        with contextlib.redirect_stdout(sys.stderr):       # call_mods prints its own timing line; stdout carries ONE JSON line
            cm.call_mods(path, "unused", out, 17, 360, batch, 0.001, 2, 1, True, True, True, True, None, engine=eng)   # warm
            t0 = time.perf_counter()
            n = cm.call_mods(path, "unused", out, 17, 360, batch, 0.001, 2, 1, True, True, True, True, None, engine=eng)
            dt = time.perf_counter() - t0
        assert n == rows and sum(1 for _ in open(out)) == rows
        size = os.path.getsize(path)
    finally:
        import shutil
        shutil.rmtree(tmpdir, ignore_errors=True)
    return {"value": round(rows / dt, 1), "unit": "sites/s", "rows": rows, "input_MB": round(size / 1e6, 1),
            "host_cores": effective_cores(),
            "engine_batch": int(getattr(eng, "max_batch", batch)),
            "path": "call_mods(feature TSV -> result TSV): native reader + ds_submit / ds_wait (whole engine batches filled across "
                    "queue items) + native formatter, 20 sites per read, f5_batch_num 50"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--windows", type=int, default=5, help="repetitions of the timed K-step window (median reported)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile-pass", action="store_true")
    ap.add_argument("--no-standalone-pass", action="store_true",
                    help="skip pass C (every launch on one stream); used for the rocprofv3 cross-check, whose per-kernel "
                         "averages would otherwise blend co-resident and stand-alone launches")
    ap.add_argument("--no-host-path", action="store_true", help="skip the pcie_inclusive and e2e_tsv legs")
    ap.add_argument("--no-fast-mode", "--no-three-step", dest="no_fast_mode", action="store_true",
                    help="skip the fast_mode_folded leg (the default engine with the folded joint model)")
    ap.add_argument("--sharded-rows-per-rank", type=int, default=40960,
                    help="N > 1: rows per rank of the e2e_tsv_sharded leg (feature TSV -> result TSV through the sharded call_mods route)")
    ap.add_argument("--no-split", action="store_true", help="skip the fp32_class_bf16x3 leg (DS_PRECISION_BF16X3 beside the headline)")
    ap.add_argument("--no-configs2", action="store_true", help="skip the BASELINE configs[2] leg (bf16 modes at batch 4096)")
    ap.add_argument("--lstm-tiling", default="auto", help="diagnostic: force a BiLSTM cell-kernel variant (Engine(lstm_tiling=...))")
    ap.add_argument("--slots", type=int, default=0, help="diagnostic: forwards in flight (0 = engine default, 8)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="process-group backend for N > 1: nccl = RCCL over xGMI (the product path); gloo = the same code with "
                         "the collectives on host tensors (tests: several ranks on the one GPU of a test box)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="tests only: ranks take GPU local_rank %% device_count (several engines on one GPU); needs --backend gloo")
    ap.add_argument("--collective-timeout", type=float, default=600.0,
                    help="seconds before a collective that a peer never joins aborts the job with a non-zero exit")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / collective / JSON plumbing only (stub engine, gloo, CPU): NOT a measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))            # fresh child ranks; this parent never touched the GPU

    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started under a launcher with WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    import datetime
    pg_timeout = datetime.timedelta(seconds=args.collective_timeout)
    if args.share_gpu and args.backend != "gloo":
        raise SystemExit("--share-gpu needs --backend gloo (RCCL wants one GPU per rank)")
    gpu_index = local_rank % max(1, torch.cuda.device_count()) if args.share_gpu else local_rank
    on_host = args.dry_run or args.backend == "gloo"          # collectives on host tensors
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if on_host:
            dist.init_process_group("gloo", timeout=pg_timeout)
        else:
            torch.cuda.set_device(gpu_index)
            dist.init_process_group("nccl", device_id=torch.device("cuda", gpu_index), timeout=pg_timeout)
    dev = torch.device("cpu") if args.dry_run else torch.device("cuda", gpu_index)
    cdev = torch.device("cpu") if on_host else dev                # where collective payloads live
    if not args.dry_run:
        torch.cuda.set_device(dev)
    dev_sync = (lambda: None) if args.dry_run else torch.cuda.synchronize

    from deepsignal_amd import spec, synth, weights as W

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ     # under torch.distributed.run, also with one rank
    if dist is None and launched and world == 1 and not args.dry_run:
        # one rank under the launcher: same code path as N > 1 (process group on RCCL, the gather inside the timed window)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if on_host:
            dist.init_process_group("gloo", timeout=pg_timeout)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", gpu_index), timeout=pg_timeout)

    if args.dry_run:
        w = None
        eng = _DryRunEngine(torch)
    else:
        from deepsignal_amd import engine as _engine_mod
        from deepsignal_amd.engine import Engine
        w = W.random_weights(seed=W.WEIGHT_SEED)          # TF-initializer style, randomised BN
        # the headline engine runs the REFERENCE GRAPH: joint model as avgpool + dense(6032, 6032) + dense(6032, 2)
        # (layers.py:233-238,257-263); the folded default engine is timed beside it (fast_mode_folded)
        eng = Engine(device=gpu_index, max_batch=BATCH, lstm_tiling=args.lstm_tiling, slots=args.slots, fold_fc=False)
        eng.load_weights(w)

    # this rank's shard: its own reads (20 sites per read), NPOOL distinct batches resident in HBM
    NPOOL = 8
    feats = synth.synthetic_features(NPOOL * BATCH, seed=synth.FEATURE_SEED + rank)
    d = {k: torch.from_numpy(feats[k]).to(dev) for k in ("kmer", "means", "stds", "sanums", "signals")}
    K, Wm = args.steps, args.warmup
    out_act = torch.zeros((max(K, 1), BATCH, 2), dtype=torch.float32, device=dev)
    out_pred = torch.zeros((max(K, 1), BATCH), dtype=torch.int32, device=dev)

    def make_step(e):
        def step(i, slot):
            b = (i % NPOOL) * BATCH
            e.run_device(BATCH, d["kmer"][b:b + BATCH].data_ptr(), d["means"][b:b + BATCH].data_ptr(),
                         d["stds"][b:b + BATCH].data_ptr(), d["sanums"][b:b + BATCH].data_ptr(),
                         d["signals"][b:b + BATCH].data_ptr(), out_act[slot].data_ptr(), out_pred[slot].data_ptr())
        return step

    gathered = {"bytes": 0}
    per_rank = []          # per timed window: every rank's elapsed seconds

    def timed_windows(e, nwin):
        """W untimed steps, then nwin windows of EXACTLY K steps, each bracketed by barrier + synchronize on both sides,
        max over ranks; with a process group the result gather to rank 0 is inside the window."""
        step = make_step(e)

        def fence():
            e.sync()
            dev_sync()
            if dist is not None:
                dist.barrier()
                dev_sync()

        for i in range(Wm):
            step(i, 0)
        wins = []
        for rep in range(max(1, nwin)):
            fence()
            t0 = time.perf_counter()
            for i in range(K):
                step(i, i)
            e.sync()
            if dist is not None:
                # the path's only exchange: gather f32[n,2] + i32[n] (12 B/site) to the writer rank over RCCL
                from deepsignal_amd import sharding
                # rank r owns the sites r, r + world, ...: the writer derives the indices, only the 12 B/site travel
                g_act, g_pred = sharding.gather_results(
                    out_act.reshape(-1, 2), out_pred.reshape(-1), None, dist, dst=0, device=cdev, as_numpy=False,
                    force_collective=True,
                    index_of_rank=lambda r, cnt: torch.arange(r, r + world * cnt, world, dtype=torch.int64, device=cdev))
                if rank == 0:
                    assert g_act.shape[0] == world * K * BATCH
                    gathered["bytes"] = int(g_act.numel() * 4 + g_pred.numel() * 4)
            fence()
            el = time.perf_counter() - t0
            if dist is not None:
                # every rank's own clock around the window; the window's time is the MAX over ranks
                mine = torch.tensor([el], dtype=torch.float64, device=cdev)
                every = torch.zeros(world, dtype=torch.float64, device=cdev)
                dist.all_gather_into_tensor(every, mine)
                per_rank.append([float(x) for x in every.cpu().tolist()])
                el = max(per_rank[-1])
            wins.append(el)
        return wins

    windows = timed_windows(eng, args.windows)
    assert bool(torch.isfinite(out_act).all())
    elapsed = sorted(windows)[len(windows) // 2]          # median window

    result = {
        "metric": "CpG sites/sec (batch=512, k=17, sig=360)",
        "value": round(world * K * BATCH / elapsed, 1) if K else 0.0,
        "unit": "sites/s", "n_gpus": world, "steps": K, "warmup": Wm,
        "ms_per_step": round(1e3 * elapsed / max(K, 1), 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": WORKLOAD, "batch": BATCH, "kmer_len": 17, "signal_len": 360,
                   "joint_model": "three-step, as the reference graph: avgpool(7) kernel, dense(6032, 6032) GEMM, dense(6032, 2) "
                                  "+ sigmoid + argmax (layers.py:233-238,257-263); the folded engine is in fast_mode_folded",
                   "sharding": "by read, %d rank(s), full weight replica per GPU" % world},
        "windows": {"n": len(windows), "steps_each": K, "statistic": "median",
                    "sites_per_s": [round(world * K * BATCH / x, 1) for x in windows] if K else [],
                    "min": round(world * K * BATCH / max(windows), 1) if K else 0.0,
                    "max": round(world * K * BATCH / min(windows), 1) if K else 0.0,
                    #  what a window is, so that a 20-step and a 400-step run of one library are not read as two results
                    "boundary": "inputs resident in HBM: ds_forward_device, K forwards issued back to back over the engine's slots (up to 8 in "
                                "flight), barrier + synchronize on both sides; a window carries ONE pipeline fill and drain (about one step): "
                                "~4 % of a 20-step window, 0.2 % of a 400-step one -- the same library reads 406 - 409 k sites/s with --steps 20 "
                                "and 415 - 418 k with the default 400 on one box (profiles/r06_item3_r04_vs_head.json). The host-inclusive "
                                "rates ride along in pcie_inclusive / ds_forward_blocking / e2e_tsv"},
    }
    if not args.dry_run and _engine_mod.LIBRARY_OVERRIDE:
        result["library_override"] = _engine_mod.LIBRARY_OVERRIDE       # DS_HIP_LIBRARY was set: NOT the in-tree product library
    if dist is not None:
        result["windows"]["per_rank_ms_per_step"] = [[round(1e3 * x / max(K, 1), 4) for x in w_] for w_ in per_rank[:len(windows)]]
        result["gather"] = {"backend": dist.get_backend(), "world": world, "bytes_per_window": gathered["bytes"],
                            "what": "f32[n,2] + i32[n] of every rank to rank 0 (sharding.gather_results), inside every timed window"}
    if args.dry_run:
        result.update({"dry_run": True, "data": "none (launcher dry run: stub engine, gloo, no forward executed)",
                       "value": 0.0, "ms_per_step": 0.0})
        result["windows"].update({"sites_per_s": [], "min": 0.0, "max": 0.0})
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(result))
        return

    step = make_step(eng)
    if rank == 0 and not args.no_profile_pass:
        # HIP events on the engine's own streams, eager replay of the same K steps.
        # pass A (mode 2, one pair per launch): per-stage breakdown
        KP = min(K, 50)                   # replayed steps per event pass
        eng.set_profiling(2)
        eng.reset_stage_times()
        for i in range(KP):
            step(i, i)
        eng.sync()
        stages = {s["name"]: round(1e3 * s["total_ms"] / max(s["calls"], 1), 1) for s in eng.stage_times()}
        exec_flops_per_site = float(sum(s["flops_per_site"] for s in eng.stage_times()))
        # pass B (mode 1, one pair per run of same-kernel launches): per-kernel durations for the roofline
        eng.set_profiling(1)
        eng.reset_stage_times()
        tp = time.perf_counter()
        for i in range(KP):
            step(i, i)
        eng.sync()
        prof_ms = 1e3 * (time.perf_counter() - tp) / max(KP, 1)
        ks = [k for k in eng.kernel_stats() if k["launches"]]
        # pass C (mode 3): the same, with every launch on one stream -> stand-alone duration of each kernel
        alone = {}
        if not args.no_standalone_pass:
            eng.set_profiling(3)
            eng.reset_stage_times()
            for i in range(KP):
                step(i, i)
            eng.sync()
            alone = {k["name"]: k for k in eng.kernel_stats() if k["launches"]}
        eng.set_profiling(0)
        # The roofline object LEADS with the stand-alone figure of the kernel that takes most of a serial step (pass C: every
        # launch on one stream, HIP events on that stream) -- the duration `rocprofv3 --kernel-trace --stats` of the serial
        # probe reproduces (profiles/rNN_serial_kernel_stats.csv, tools/collect_profiles.sh). The co-resident figure (pass
        # B: the kernels of the two model halves and of up to 8 forwards share the GPU, as in the timed region) is a
        # property of the mix and rides along under "co_resident".
        frac_of = lambda k: k["flops"] / (k["total_ms"] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS if k["total_ms"] else 0.0
        dom_co = max(ks, key=lambda k: k["total_ms"])
        dom = max(alone.values(), key=lambda k: k["total_ms"]) if alone else dom_co
        per_launch_flops = dom["flops"] / dom["launches"]
        avg_ms = dom["total_ms"] / dom["launches"]
        achieved = per_launch_flops / (avg_ms * 1e-3) / 1e12
        result["roofline"] = {
            "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None,
            "kernel": dom["name"], "launches_per_step": dom["launches"] // max(KP, 1),
            "avg_launch_us": round(avg_ms * 1e3, 2), "flops_per_launch": per_launch_flops,
            "measured": ("stand-alone: every launch of the step on ONE stream, HIP events on that stream, eager replay of %d steps; "
                         "the kernel with the largest share of the serial step" % KP) if alone else "co-resident (pass C skipped)",
            "profiled_ms_per_step": round(prof_ms, 4),
            # FLOPs the engine EXECUTES per site (layer-0 input projection = table lookup) next to the reference
            # graph's contract FLOPs; whole_path_tflops prices the executed ones over the TIMED region (value)
            "whole_path_tflops": round(exec_flops_per_site * result["value"] / world / 1e12, 2),
            "whole_path_frac": round(exec_flops_per_site * result["value"] / world / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
            "flops_per_site": {"executed": exec_flops_per_site, "reference_graph": spec.FLOPS_PER_SITE},
        }
        if alone:
            result["roofline"]["standalone_fracs"] = {n_: round(frac_of(k), 4) for n_, k in alone.items() if k["flops"]}
            co_ms = dom_co["total_ms"] / dom_co["launches"]
            co_tf = dom_co["flops"] / dom_co["launches"] / (co_ms * 1e-3) / 1e12
            result["roofline"]["co_resident"] = {
                "kernel": dom_co["name"], "avg_launch_us": round(co_ms * 1e3, 2), "achieved": round(co_tf, 2),
                "frac": round(co_tf / PEAK_FP32_MFMA_TFLOPS, 4),
                "note": "the kernel with the largest summed duration while the signal-model and event-model kernels of the "
                        "forwards in flight share the CUs (how the timed region runs); its launches wait for CUs other kernels "
                        "own, so this is a figure of the mix, not of the kernel"}
        result["roofline"].update(pmc_traffic(dom["name"]))
        # the two figures the north star names: MFMA utilisation on the FC path, HBM GB/s on the conv path
        src = alone if alone else {k["name"]: k for k in ks}
        fc = [k for n_, k in src.items() if n_.startswith("gemm_kernel<1,3,4,1")]        # dense(6032, 6032) of the joint model
        if fc:
            f_ms = fc[0]["total_ms"] / fc[0]["launches"]
            f_tf = fc[0]["flops"] / fc[0]["launches"] / (f_ms * 1e-3) / 1e12
            result["roofline"]["fc_path"] = dict(
                {"kernel": fc[0]["name"], "avg_launch_us": round(f_ms * 1e3, 2), "achieved": round(f_tf, 2), "unit": "TFLOP/s",
                 "mfma_utilisation": round(f_tf / PEAK_FP32_MFMA_TFLOPS, 4),
                 "how": "2 * n * 6032 * 6032 FLOP per launch / HIP-event duration (%s) / fp32 MFMA peak"
                        % ("every launch on one stream" if alone else "co-resident")}, **pmc_mfma_busy(fc[0]["name"]))
        conv = [k for n_, k in src.items() if n_.startswith(("stem1_kernel", "stem23_kernel", "inception_fused_kernel", "avgpool7_kernel"))]
        if conv:
            c_ms = sum(k["total_ms"] for k in conv) / max(KP, 1)
            c_bytes = CONV_BYTES_PER_SITE_MODULE_GRANULAR * BATCH
            result["roofline"]["conv_path"] = {
                "kernels": [k["name"] for k in conv], "us_per_step": round(c_ms * 1e3, 1),
                "hbm_GBps": round(c_bytes / (c_ms * 1e-3) / 1e9, 1), "hbm_frac_of_8TBps": round(c_bytes / (c_ms * 1e-3) / 8e12, 4),
                "tflops": round(sum(k["flops"] for k in conv) / max(KP, 1) / (c_ms * 1e-3) / 1e12, 2),
                "how": "module-granular algorithmic bytes (SURVEY.md 8d: 1,651,680 B/site, every stem conv / pool / module reads "
                       "its input and writes its output once) / summed HIP-event durations; fp32 modules are MFMA-bound, the "
                       "HBM-bound form of the conv path is the bf16 mode (DESIGN.md section 9)"}
        result["kernels"] = {k["name"]: {"launches_per_step": k["launches"] // max(KP, 1),
                                          "us_per_step": round(1e3 * k["total_ms"] / max(KP, 1), 1),
                                          "tflops": round(k["flops"] / (k["total_ms"] * 1e-3) / 1e12, 2) if k["total_ms"] else 0,
                                          "us_per_step_alone": round(1e3 * alone[k["name"]]["total_ms"] / max(KP, 1), 1)
                                          if k["name"] in alone else None}
                             for k in ks}
        result["stages_us_per_step"] = stages
    solo = rank == 0 and world == 1 and not launched
    if solo and not args.no_host_path:
        result["pcie_inclusive"] = pcie_inclusive(eng, feats, max(K, 64), BATCH)
        result["ds_forward_blocking"] = forward_blocking(eng, feats, BATCH)
    if solo and not args.no_fast_mode:
        # the product's default engine: joint model folded into one 6032 x 2 matrix at weight load (float64 product;
        # layers.py:75-77,233-238,257-263 have no bias / activation) -- same inputs, same timed-region rules, 3 windows
        engf = Engine(device=local_rank, max_batch=BATCH, lstm_tiling=args.lstm_tiling, slots=args.slots)
        engf.load_weights(w)
        wf = timed_windows(engf, 3)
        elf = sorted(wf)[len(wf) // 2]
        stf = {s_["name"]: s_ for s_ in engf.stage_times()}
        fm = {"value": round(K * BATCH / elf, 1) if K else 0.0, "unit": "sites/s", "ms_per_step": round(1e3 * elf / max(K, 1), 4),
              "executed_flops_per_site": float(sum(s_["flops_per_site"] for s_ in stf.values())),
              "note": "Engine() default: avgpool x dense(6032, 6032) x dense(6032, 2) -> one 6032 x 2 matrix; parity-tested against "
                      "the three-step form and the oracle (tests/test_gpu_parity.py); NOT the headline (SURVEY.md K13)"}
        fm["whole_path_tflops"] = round(fm["executed_flops_per_site"] * fm["value"] / 1e12, 2)
        if not args.no_host_path:
            fm["pcie_inclusive"] = pcie_inclusive(engf, feats, max(K, 64), BATCH)
            fm["ds_forward_blocking"] = forward_blocking(engf, feats, BATCH)
            # end to end as the CLI runs it: call_mods creates its engine for ENGINE_BATCH sites per forward whatever --batch_size
            # says (a site's result does not depend on its batch mates) and fills whole engine batches across queue items
            from deepsignal_amd import call_modifications as _cm
            enge = Engine(device=gpu_index, max_batch=_cm.ENGINE_BATCH["fp32"])
            enge.load_weights(w)
            fm["e2e_tsv"] = e2e_tsv(enge, feats, 163840, BATCH)
            enge.close()
            result["e2e_tsv"] = dict(fm["e2e_tsv"], engine="default (folded joint model), created as `deepsignal call_mods` creates it")
        result["fast_mode_folded"] = fm
        engf.close()
    eng.close()
    if dist is not None and not args.no_host_path:
        # the product's multi-GPU route: every rank takes part (collectives inside), rank 0 reports. Also with ONE rank under the
        # launcher, so that the leg's RCCL calls have run on a GPU box before a multi-GPU node ever sees them (tests/test_gpu_rccl.py).
        # A failure here must not take the headline down with it: it is caught on the rank it happens on and reported as text (a
        # rank that fails inside a collective still ends the job through the process group's timeout)
        try:
            shard = e2e_tsv_sharded(Engine, w, feats, dist, rank, world, gpu_index, cdev, args.sharded_rows_per_rank, BATCH)
        except Exception as exc:
            shard = {"error": repr(exc)}
        if rank == 0:
            result["e2e_tsv_sharded"] = shard
    if solo and not args.no_split:
        result["fp32_class_bf16x3"] = bf16x3_leg(Engine, w, torch, make_step, timed_windows, local_rank, args.slots, args.lstm_tiling, K,
                                                 host_feats=None if args.no_host_path else feats)
    if solo and not args.no_configs2:
        result["configs2_bf16_batch4096"] = configs2_leg(Engine, w, torch, dev, local_rank, host_path=not args.no_host_path)
        if "e2e_tsv" in result["configs2_bf16_batch4096"].get("bf16_all", {}):
            result["e2e_tsv_bf16_all"] = dict(result["configs2_bf16_batch4096"]["bf16_all"]["e2e_tsv"], engine="bf16_all, max_batch 4096 (`deepsignal call_mods --precision bf16_all`)")
    if solo and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(w)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
