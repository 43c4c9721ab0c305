"""Native feature-TSV reader alone: sites/s vs parser threads (host only)."""
import os, sys, time, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np
from deepsignal_amd import synth, fastio
from deepsignal_amd.utils.process_utils import code2base_dna
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
feats = synth.synthetic_features(n, seed=1)
path = os.path.join(tempfile.gettempdir(), "rd_%d.tsv" % n)
with open(path, "w") as f:
    for i in range(n):
        f.write("\t".join(["chr1", str(i), "+", str(i), "read%d" % (i // 20), "t",
                           "".join(code2base_dna[int(c)] for c in feats["kmer"][i]),
                           ",".join("%.6f" % x for x in feats["means"][i]), ",".join("%.6f" % x for x in feats["stds"][i]),
                           ",".join(str(int(x)) for x in feats["sanums"][i]), ",".join("%.6f" % x for x in feats["signals"][i]), "1"]) + "\n")
print("file MB", os.path.getsize(path) / 1e6)
for th in (0, 4, 8, 16, 32, 64):
    r = fastio.FeatureReader(path, 17, 360, nthreads=th)
    t0 = time.perf_counter(); m = 0
    for it in r.items(400): m += len(it.labels)
    dt = time.perf_counter() - t0
    r.close()
    print("threads %3d: %.0f sites/s (%.0f MB/s)" % (th, m / dt, os.path.getsize(path) / 1e6 / dt))
