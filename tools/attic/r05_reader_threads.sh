# feature TSV -> result TSV rate against the native reader's thread count (0 = the library's default: usable CPUs), three runs each
for prec in bf16x3 bf16_all fp32; do
python3 - $prec <<'PY'
import os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import numpy as np
from deepsignal_amd import call_modifications as cm, synth, weights as W, fastio
from deepsignal_amd.engine import Engine
from deepsignal_amd.utils.process_utils import code2base_dna
prec = sys.argv[1]
rows = 327680 if prec != "fp32" else 163840
feats = synth.synthetic_features(4096, seed=1)
tmp = tempfile.mkdtemp(prefix="ds_e2e_"); path = os.path.join(tmp, "features.tsv")
tails = ["\t".join(["".join(code2base_dna[int(c)] for c in feats["kmer"][i]), ",".join("%.6f" % x for x in feats["means"][i]),
                    ",".join("%.6f" % x for x in feats["stds"][i]), ",".join(str(int(x)) for x in feats["sanums"][i]),
                    ",".join("%.6f" % x for x in feats["signals"][i]), "1"]) for i in range(4096)]
with open(path, "w") as f:
    for i in range(rows):
        f.write("chr1\t%d\t+\t%d\tread_%06d\tt\t%s\n" % (1000 + i, i, i // 20, tails[i % 4096]))
B = cm.engine_batch_for(512, prec)
eng = Engine(max_batch=B, precision=prec); eng.load_weights(W.random_weights(seed=1))
args = (path, "x", os.path.join(tmp, "out.tsv"), 17, 360, 512, 0.001, 2, 1, True, True, True, True, None)
orig = fastio.FeatureReader.__init__
out = []
for nt in (0, 16, 15, 14, 12, 8):
    def init(self, path, kmer_len=17, signal_len=360, nthreads=0, _nt=nt): orig(self, path, kmer_len, signal_len, _nt)
    fastio.FeatureReader.__init__ = init
    cm.call_mods(*args, engine=eng)
    r = []
    for _ in range(3):
        t0 = time.perf_counter(); cm.call_mods(*args, engine=eng); r.append(rows / (time.perf_counter() - t0))
    out.append((nt, round(sorted(r)[1])))
print(prec, "engine batch", B, os.cpu_count(), len(os.sched_getaffinity(0)), out)
import shutil; shutil.rmtree(tmp, ignore_errors=True)
PY
done
