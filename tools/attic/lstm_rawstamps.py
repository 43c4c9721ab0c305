"""Per-workgroup stamps of one BiLSTM diagonal: distribution of loop times, by XCD / CU / logical tile."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from deepsignal_amd import synth, weights as W
from deepsignal_amd.engine import Engine
var = sys.argv[1] if len(sys.argv) > 1 else "lds1"
diag = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B = 512
e = Engine(max_batch=B, slots=1, serial=True, debug_stamps=True, lstm_tiling=var); e.load_weights(W.random_weights(seed=1))
e.set_graph(False)
f = synth.synthetic_features(B, seed=2)
args = [f[k] for k in ("kmer", "means", "stds", "sanums", "signals")]
for _ in range(3): e.run(*args)
e.intermediate("lstm_rawstamps%d" % diag, (1024, 8))
e.run(*args)
s = e.intermediate("lstm_rawstamps%d" % diag, (1024, 8))
e.close()
v = s[:, 7] > 0
n = int(v.sum())
print("variant", var, "diag", diag, "wgs", n)
blk = np.arange(1024)[v]; s = s[v]
entry, pro, loop, ex, end, key = s[:, 0] * 0.01, s[:, 1], s[:, 2], s[:, 3], s[:, 4] * 0.01, s[:, 5].astype(int)
print("entry us: min %.2f max %.2f | prologue cyc pct [10,50,90,100]: %s | loop cyc pct: %s | end us pct: %s" % (
    entry.min(), entry.max(), np.percentile(pro, [10, 50, 90, 100]).astype(int), np.percentile(loop, [0, 10, 50, 90, 100]).astype(int),
    np.round(np.percentile(end, [10, 50, 90, 100]), 1)))
xcc = key >> 8
for x in range(8):
    m = xcc == x
    if m.any():
        print("  xcc %d: wgs %3d  prologue med %6d  loop med %6d max %6d  end max %.1f us  blockIdx%%8 %s" % (
            x, m.sum(), np.median(pro[m]), np.median(loop[m]), loop[m].max(), end[m].max(), sorted(set((blk[m] % 8).tolist()))))
# per CU: the three workgroups' loop times
cu = {}
for k, l, en, st, b in zip(key, loop, end, entry, blk): cu.setdefault(k, []).append((int(b) >> 3, round(float(st), 2), int(l), round(float(en), 1)))
ks = sorted(cu)[:12] + sorted(cu)[40:44] + sorted(cu)[-3:]
print("  (per CU: XCD-local workgroup index = blockIdx >> 3, entry us, loop cycles, exit us)")
for k in ks: print("  cu %04x:" % k, cu[k])
# by logical tile position: remap as the kernel does
total = n; q, r = total >> 3, total & 7
bid = (blk & 7) * q + np.minimum(blk & 7, r) + (blk >> 3)
order = np.argsort(bid)
lo = loop[order]
print("loop cycles by logical tile, means of 48-tile runs:", [int(lo[i:i + 48].mean()) for i in range(0, n, 48)])
