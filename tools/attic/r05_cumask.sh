# EXPERIMENT (its engine switch, ds_config.reserved[2] bits 8..11 -> hipExtStreamCreateWithCUMask for the two branch streams, is NOT in the
# library: re-add it to run this): the event-model stream on a share of the CUs (eighths), the signal-model stream on the rest (hipExtStreamCreateWithCUMask,
# eager issue), against the default (no masks, captured graphs) and against eager issue without masks
python - <<'PY' 2>/dev/null
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from deepsignal_amd import synth, weights as W
from deepsignal_amd.engine import Engine
B = 512; dev = torch.device("cuda", 0)
w = W.random_weights(seed=W.WEIGHT_SEED)
f = synth.synthetic_features(8 * B, seed=1)
keys = ("kmer", "means", "stds", "sanums", "signals")
d = {k: torch.from_numpy(f[k]).to(dev) for k in keys}
act = torch.zeros((B, 2), dtype=torch.float32, device=dev); pred = torch.zeros((B,), dtype=torch.int32, device=dev)
def rate(e, steps=300):
    def step(i):
        b = (i % 8) * B
        e.run_device(B, *(d[k][b:b + B].data_ptr() for k in keys), act.data_ptr(), pred.data_ptr())
    for i in range(30): step(i)
    e.sync(); r = []
    for _ in range(5):
        t0 = time.perf_counter()
        for i in range(steps): step(i)
        e.sync(); r.append(steps * B / (time.perf_counter() - t0))
    return round(sorted(r)[2])
for prec in ("bf16x3", "fp32"):
    for fold in (True, False):
        out = []
        for share in (0, -1, 2, 3, 4, 5):
            e = Engine(max_batch=B, precision=prec, fold_fc=fold, event_cu_eighths=max(share, 0)); e.load_weights(w)
            if share == -1: e.set_graph(False)
            out.append((share, rate(e))); e.close()
        print(prec, "fold" if fold else "3step", out, flush=True)
PY
