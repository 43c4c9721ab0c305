"""Host-side I/O throughput (scope row f1): Python reader/formatter (reference algorithm) vs native."""
import os, sys, time, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np
from deepsignal_amd import call_modifications as cm, fastio, synth
from deepsignal_amd.utils.process_utils import code2base_dna

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
feats = synth.synthetic_features(n, seed=1)
path = os.path.join(tempfile.gettempdir(), "io_bench_%d.tsv" % n)
with open(path, "w") as f:
    for i in range(n):
        f.write("\t".join(["chr1", str(i), "+", str(i), "read%d" % (i // 20), "t",
                           "".join(code2base_dna[int(c)] for c in feats["kmer"][i]),
                           ",".join("%.6f" % x for x in feats["means"][i]), ",".join("%.6f" % x for x in feats["stds"][i]),
                           ",".join(str(int(x)) for x in feats["sanums"][i]), ",".join("%.6f" % x for x in feats["signals"][i]), "1"]) + "\n")
mb = os.path.getsize(path) / 1e6


class NullEngine:
    def run(self, kmer, *a):
        k = len(kmer)
        act = np.full((k, 2), 0.25, np.float32); act[:, 1] = 0.6
        return act, np.ones(k, np.int32)

t0 = time.perf_counter(); cm.call_mods(path, "x", path + ".py.out", 17, 360, 512, 0.001, 2, 1, False, True, True, True, None, engine=NullEngine(), native_io=False); tp = time.perf_counter() - t0
t0 = time.perf_counter(); cm.call_mods(path, "x", path + ".nat.out", 17, 360, 512, 0.001, 2, 1, False, True, True, True, None, engine=NullEngine(), native_io=True); tn = time.perf_counter() - t0
assert open(path + ".py.out").read() == open(path + ".nat.out").read()
print("file %.1f MB, %d rows, cores %d" % (mb, n, os.cpu_count()))
print("python reader+formatter: %.2f s = %.0f sites/s (%.1f MB/s)" % (tp, n / tp, mb / tp))
print("native reader+formatter: %.2f s = %.0f sites/s (%.1f MB/s)  speed-up %.1fx" % (tn, n / tn, mb / tn, tp / tn))
