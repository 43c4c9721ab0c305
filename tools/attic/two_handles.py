import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from deepsignal_amd import synth, weights as W
from deepsignal_amd.engine import Engine
w = W.random_weights(seed=W.WEIGHT_SEED)
NH = int(sys.argv[1]) if len(sys.argv) > 1 else 2
engs = []
for i in range(NH):
    e = Engine(device=0, max_batch=512); e.load_weights(w); engs.append(e)
dev = torch.device('cuda', 0)
feats = synth.synthetic_features(8 * 512, seed=1)
d = {k: torch.from_numpy(feats[k]).to(dev) for k in ("kmer", "means", "stds", "sanums", "signals")}
K = 60
out_act = torch.zeros((K, 512, 2), dtype=torch.float32, device=dev); out_pred = torch.zeros((K, 512), dtype=torch.int32, device=dev)
def step(i):
    b = (i % 8) * 512; e = engs[i % NH]
    e.run_device(512, d["kmer"][b:b+512].data_ptr(), d["means"][b:b+512].data_ptr(), d["stds"][b:b+512].data_ptr(), d["sanums"][b:b+512].data_ptr(), d["signals"][b:b+512].data_ptr(), out_act[i].data_ptr(), out_pred[i].data_ptr())
for i in range(6): step(i)
for e in engs: e.sync()
t0 = time.perf_counter()
for i in range(K): step(i)
for e in engs: e.sync()
dt = time.perf_counter() - t0
print("handles", NH, "ms/step %.4f" % (1e3 * dt / K), "sites/s %.0f" % (K * 512 / dt))
