"""bench.py's launch contract on a CPU-only box: `python bench.py --gpus N` with no launcher environment must start its
own ranks (fresh child processes, spawned before the parent touches the GPU) and print ONE JSON line. `--dry-run` swaps
the engine for a stub and RCCL for gloo, so what runs here is the launcher, the barriers, the result gather and the
JSON plumbing -- not a measurement (the line says so)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", *flags], env=env, cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout.decode()
    return json.loads(lines[0])


def test_gpus_2_self_launches_two_ranks():
    r = _run("--gpus", "2", "--steps", "20", "--warmup", "5")
    assert r["n_gpus"] == 2 and r["steps"] == 20 and r["warmup"] == 5 and r["dry_run"] is True
    assert r["scaling"] == "weak" and r["unit"] == "sites/s" and r["windows"]["n"] == 5
    assert "2 rank(s)" in r["config"]["sharding"]


def test_gpus_1_line_has_the_contract_keys():
    r = _run("--steps", "4", "--warmup", "1", "--windows", "3")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config"):
        assert key in r
    assert r["n_gpus"] == 1 and r["windows"]["n"] == 3 and r["vs_baseline"] is None


def test_committed_pmc_constants_resolve_for_the_kernels_the_line_quotes():
    """bench.py cannot read counters from inside the process: roofline.traffic, fc_path.mfma_busy_pmc and
    conv_path_hbm.traffic_per_step are constants of the latest profiles/rNN_pmc_*.json. A renamed kernel or file would
    silently turn them into null: the three lookups of the default line must resolve, name the file and the commit, and
    be consistent with the raw counter CSVs' summaries."""
    sys.path.insert(0, ROOT)
    import bench
    t = bench.pmc_traffic("lstm_cell_lds_kernel<1>")
    assert t["traffic"] and abs(t["traffic"] - t["traffic_fetch"] - t["traffic_write"]) <= 2
    assert "NOT measured in this run" in t["traffic_source"] and ("collected at commit" in t["traffic_source"] or "no build identity" in t["traffic_source"])
    assert 15e6 < t["traffic"] < 60e6                      # ~28 MB per BiLSTM diagonal at 512 sites
    f = bench.pmc_traffic("inception_fused_bf16_kernel<3>", "r[0-9][0-9]_bf16_all_4096_pmc_traffic.json")
    assert f["traffic"] and "bf16_all_4096_pmc_traffic.json" in f["traffic_source"]
    assert 3 * f["traffic"] < 992 * 564 * 4096             # the LDS chain moves fewer bytes than the module-granular figure
    m = bench.pmc_mfma_busy("gemm_kernel<1,3,4,1,0,2,2,1>")
    assert 0.5 < m["mfma_busy_pmc"] < 1.0 and "NOT measured in this run" in m["mfma_busy_source"]
