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
