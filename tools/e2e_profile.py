"""Diagnostic: where the host side of call_mods(feature TSV -> result TSV) spends its time (main thread, cProfile)."""
import cProfile, os, pstats, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import numpy as np
from deepsignal_amd import call_modifications as cm, synth, weights as W
from deepsignal_amd.engine import Engine
from deepsignal_amd.utils.process_utils import code2base_dna
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 163840
feats = synth.synthetic_features(4096, seed=1)
tmp = tempfile.mkdtemp(prefix="ds_e2e_")
path = os.path.join(tmp, "features.tsv")
tails = ["\t".join(["".join(code2base_dna[int(c)] for c in feats["kmer"][i]), ",".join("%.6f" % x for x in feats["means"][i]),
                    ",".join("%.6f" % x for x in feats["stds"][i]), ",".join(str(int(x)) for x in feats["sanums"][i]),
                    ",".join("%.6f" % x for x in feats["signals"][i]), "1"]) for i in range(4096)]
with open(path, "w") as f:
    for i in range(rows):
        f.write("chr1\t%d\t+\t%d\tread_%06d\tt\t%s\n" % (1000 + i, i, i // 20, tails[i % 4096]))
eng = Engine(max_batch=int(os.environ.get("E2E_BATCH", "512")), slots=int(os.environ.get("E2E_SLOTS", "0")), precision=os.environ.get("E2E_PRECISION", "fp32")); eng.load_weights(W.random_weights(seed=1))
args = (path, "x", os.path.join(tmp, "out.tsv"), 17, 360, 512, 0.001, 2, 1, True, True, True, True, None)
cm.call_mods(*args, engine=eng)
t0 = time.perf_counter(); cm.call_mods(*args, engine=eng); dt = time.perf_counter() - t0
print("%d rows: %.3f s = %.0f sites/s" % (rows, dt, rows / dt))
pr = cProfile.Profile(); pr.enable(); cm.call_mods(*args, engine=eng); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
# parser thread count vs end-to-end rate (the box gives 16 CPUs of quota: parser threads + this thread + the helper)
from deepsignal_amd import fastio
orig = fastio.FeatureReader.__init__
for nt in (4, 6, 8, 10, 12, 14, 16):
    def init(self, path, kmer_len=17, signal_len=360, nthreads=0, _nt=nt): orig(self, path, kmer_len, signal_len, _nt)
    fastio.FeatureReader.__init__ = init
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); cm.call_mods(*args, engine=eng); best = min(best, time.perf_counter() - t0)
    print("parser threads %2d: %.0f sites/s" % (nt, rows / best))
# the same pipeline fed from items already parsed (no parser threads running): is the parser in the GPU's way?
fastio.FeatureReader.__init__ = orig
rd = fastio.FeatureReader(path, 17, 360)
items = list(rd.items(50))
for label, sink in (("format + discard", lambda tag, data: None),):
    best = 1e9
    for _ in range(3):
        pipe = cm._RowPipeline(eng, 512, sink)
        t0 = time.perf_counter()
        for it in items:
            pipe.feed(it)
        pipe.flush()
        best = min(best, time.perf_counter() - t0)
    print("pre-parsed items, %s: %.0f sites/s" % (label, rows / best))
# submit / wait only (no formatting), same arrays every time
arrs = tuple(np.concatenate([getattr(it, k) for it in items[:2]])[:512] for k in ("kmer", "means", "stds", "lens", "signals"))
import collections
for depth in (min(8, eng.slots),):      # (engines of large forwards have four slots)
    q = collections.deque(); t0 = time.perf_counter()
    for i in range(320):
        if len(q) >= depth: eng.wait(q.popleft())
        q.append(eng.submit(*arrs))
    while q: eng.wait(q.popleft())
    print("submit / wait only, depth %d: %.0f sites/s" % (depth, 320 * 512 / (time.perf_counter() - t0)))
import shutil; shutil.rmtree(tmp, ignore_errors=True)
