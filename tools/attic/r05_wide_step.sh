# bf16x3 three-step pipelined rate with the 256 x 192 split-K dense tile on / off (same library, same box)
for round in 1 2 3; do for w in 0 1; do
python - $w <<'PY' 2>/dev/null
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
for slots in (8, 1):
    e = Engine(max_batch=B, precision="bf16x3", fold_fc=False, slots=slots, split_dense_narrow=(sys.argv[1] == "0")); e.load_weights(w)
    out.append((slots, rate(e))); e.close()
print("wide", sys.argv[1], out)
PY
done; done
