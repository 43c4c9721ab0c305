# embedding-table rows in accumulator order: kernel times (fp32, bf16x3) and pipelined rates, same box, new against lib_new (the commit before)
bash tools/ab.sh "fp32 512 10 5 lstm" new tab 2>&1 | grep -v -e Warning -e amdgpu.ids
bash tools/ab.sh "bf16x3 512 10 5 lstm" new tab 2>&1 | grep -v -e Warning -e amdgpu.ids
for round in 1 2; do for lib in new tab; do
DS_HIP_LIBRARY=$PWD/build/variants/lib_$lib.so python - $lib <<'PY' 2>/dev/null
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
for prec in ("fp32", "bf16x3"):
    for fold in (False, True):
        e = Engine(max_batch=B, precision=prec, fold_fc=fold); e.load_weights(w)
        out.append((prec, "fold" if fold else "3step", rate(e))); e.close()
print(sys.argv[1], out)
PY
done; done
