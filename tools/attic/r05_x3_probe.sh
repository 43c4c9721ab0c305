# bf16x3 in the pipelined step: slots and BiLSTM tile (three-step and folded), 3 windows of 200 steps
python - <<'PY'
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
def rate(e, steps=200):
    def step(i):
        b = (i % 8) * B
        e.run_device(B, *(d[k][b:b + B].data_ptr() for k in keys), act.data_ptr(), pred.data_ptr())
    for i in range(20): step(i)
    e.sync(); r = []
    for _ in range(3):
        t0 = time.perf_counter()
        for i in range(steps): step(i)
        e.sync(); r.append(steps * B / (time.perf_counter() - t0))
    return round(sorted(r)[1])
for fold in (False, True):
    for slots in (4, 6, 8, 12, 16):
        for til in ("auto",) if slots != 8 else ("auto", "lds1", "wide"):
            e = Engine(max_batch=B, precision="bf16x3", fold_fc=fold, slots=slots, lstm_tiling=til); e.load_weights(w)
            print("fold" if fold else "three-step", "slots", slots, til, rate(e)); e.close()
PY
