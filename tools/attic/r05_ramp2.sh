# bf16x3: few slots (1..4) against 8, with and without the narrow tile on the 2-cell ramp diagonals
for round in 1 2; do
for rc in 0 2 4; do
DS_SPLIT_RAMP_CELLS=$rc python - $rc <<'PY'
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
out = []
for prec in ("bf16x3", "fp32"):
  for fold in (False, True):
    for slots in (1, 2, 3, 4, 8):
        e = Engine(max_batch=B, precision=prec, fold_fc=fold, slots=slots); e.load_weights(w)
        out.append((prec, "fold" if fold else "3step", slots, rate(e))); e.close()
  if sys.argv[1] != "0": break
print("ramp_cells", sys.argv[1], out)
PY
done; done
