"""Golden vectors for scope row f4 by RUNNING the reference script
(/root/reference/scripts/call_modification_frequency.py — stdlib only, runs under any Python):
    python tests/golden/make_frequency_golden.py
Commits a synthetic call_mods result file and the reference's outputs for four flag combinations."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/scripts/call_modification_frequency.py"
rng = np.random.default_rng(5)
rows = []
for i in range(400):
    chrom = "chr%d" % rng.integers(1, 4)
    pos = int(rng.integers(0, 40))
    p1 = np.float32(rng.uniform(0, 1))
    p0 = np.float32(1) - p1
    rows.append("\t".join([chrom, str(pos), "+-"[pos % 2], str(1000 - pos), "read%d" % (i // 8), "t",
                           str(p0), str(p1), str(int(p1 > p0)), "ACGTACGTCGACGTACG"]))
cases = []
with tempfile.TemporaryDirectory() as d:
    inp = os.path.join(d, "calls.tsv")
    with open(inp, "w") as f:
        f.write("\n".join(rows) + "\n")
    for flags in ([], ["--sort"], ["--sort", "--bed"], ["--sort", "--prob_cf", "0.4"]):
        out = os.path.join(d, "out.tsv")
        subprocess.check_call([sys.executable, "-B", REF, "-i", inp, "-o", out] + flags, stdout=subprocess.DEVNULL,
                              cwd=os.path.dirname(REF), env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
        cases.append({"flags": flags, "output": open(out).read().splitlines()})
with open(os.path.join(HERE, "frequency_golden.json"), "w") as f:
    json.dump({"generator": "tests/golden/make_frequency_golden.py (reference script run here)", "input_rows": rows, "cases": cases}, f)
print("wrote frequency_golden.json", [(c["flags"], len(c["output"])) for c in cases])
