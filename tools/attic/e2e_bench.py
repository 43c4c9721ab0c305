"""End-to-end `call_mods` (feature TSV -> result TSV) on the GPU: Python I/O vs native I/O."""
import os, sys, time, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np
from deepsignal_amd import call_modifications as cm, synth, weights as W
from deepsignal_amd.engine import Engine
from deepsignal_amd.utils.process_utils import code2base_dna
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
precisions = sys.argv[2].split(",") if len(sys.argv) > 2 else ["fp32"]
feats = synth.synthetic_features(n, seed=1)
path = os.path.join(tempfile.gettempdir(), "e2e_%d.tsv" % n)
with open(path, "w") as f:
    for i in range(n):
        f.write("\t".join(["chr1", str(i), "+", str(i), "read%d" % (i // 20), "t",
                           "".join(code2base_dna[int(c)] for c in feats["kmer"][i]),
                           ",".join("%.6f" % x for x in feats["means"][i]), ",".join("%.6f" % x for x in feats["stds"][i]),
                           ",".join(str(int(x)) for x in feats["sanums"][i]), ",".join("%.6f" % x for x in feats["signals"][i]), "1"]) + "\n")
w = W.random_weights(seed=1)
for prec in precisions:
    eng = Engine(max_batch=512, precision=prec); eng.load_weights(w)
    for native in ((False, True) if prec == "fp32" and n <= 40000 else (True,)):
        for f5 in (50, 400):
            t0 = time.perf_counter()
            cm.call_mods(path, "x", path + ".out%d" % native, 17, 360, 512, 0.001, 2, 1, True, True, True, True, None, engine=eng, native_io=native, f5_batch_num=f5)
            dt = time.perf_counter() - t0
            print("%s native_io=%s f5_batch_num=%d: %.2f s, %.0f sites/s end to end (usable cores 16 of %d)" % (prec, native, f5, dt, n / dt, os.cpu_count()))
    if prec == "fp32" and n <= 40000:
        a = open(path + ".out0").read().split("\n"); b = open(path + ".out1").read().split("\n")
        assert a == b, "native and python outputs differ"
        print("outputs identical:", len(a) - 1, "rows")
    eng.close()
