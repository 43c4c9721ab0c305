import numpy as np, sys
sys.path.insert(0, '.')
from deepsignal_amd import weights, synth, spec
from deepsignal_amd.engine import Engine
w = weights.random_weights(seed=3, lstm_bias_std=0.05)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
feats = synth.synthetic_features(N, seed=4096)
args = [feats[k] for k in ("kmer", "means", "stds", "sanums", "signals")]
e = Engine(max_batch=N, slots=1, precision=prec, debug=True); e.load_weights(w)
d = spec.net_dims()
names = ["stem_pool", "stem_conv2", "stem_conv3"] + ["module%d" % m for m in range(1, 12)] + ["signal_feat", "joint", "fc1", "logits"] + ["lstm_%s_l%d" % (dr, l) for dr in ("fw", "bw") for l in range(3)]
def shapes(n):
    sh = {"stem_pool": (n, d.w_a, 64), "stem_conv2": (n, d.w_a, 128), "stem_conv3": (n, d.w_a, 256), "signal_feat": (n, d.signal_feat), "joint": (n, d.joint), "fc1": (n, d.joint), "logits": (n, 2)}
    for m in range(1, 12): sh["module%d" % m] = (n, d.module_width(m), 240)
    for dr in ("fw", "bw"):
        for l in range(3): sh["lstm_%s_l%d" % (dr, l)] = (n, 17, 256)
    return sh
def run(lo, hi):
    a, p = e.run(*(x[lo:hi] for x in args))
    taps = {k: e.intermediate(k, shapes(hi - lo)[k]) for k in names}
    return a, taps
a1, t1 = run(0, N)
a2, t2 = run(0, N)
print("repeat full: act diff", np.abs(a1 - a2).max(), {k: float(np.abs(t1[k] - t2[k]).max()) for k in names if np.abs(t1[k] - t2[k]).max() > 0})
lo, hi = 100, 177
a3, t3 = run(lo, hi)
print("sub vs full: act diff", np.abs(a3 - a1[lo:hi]).max())
for k in names:
    dd = np.abs(t3[k] - t1[k][lo:hi])
    if dd.max() > 0:
        idx = np.argwhere(dd > 0)
        print(k, "max", dd.max(), "count", len(idx), "first", idx[:5].tolist())
