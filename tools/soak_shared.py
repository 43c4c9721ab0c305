"""Soak of the shipped BiLSTM cell kernels in the configuration that broke round 4's persistent-BiLSTM experiment (DESIGN.md
section 9): eager issue (no captured graphs), EVERY slot's BiLSTM launches on ONE shared stream, 2 or 4 slots
of one handle in flight so that the other slots' signal-model kernels co-run with the LDS-DMA cell kernels, IDENTICAL data per slot
(every forward of the run sees the same sites: all global memory the cells read or write already holds its final value), every
forward compared bit for bit with a quiet reference pass of a default engine.

usage: python tools/soak_shared.py <precision> <batch> <slots> <min_sites> [max_seconds]
Exit status 1 on any mismatch. One JSON line per run on stdout."""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from deepsignal_amd import synth, weights as W
from deepsignal_amd.engine import Engine

prec, B, slots, min_sites = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
max_s = float(sys.argv[5]) if len(sys.argv) > 5 else 600.0
keys = ("kmer", "means", "stds", "sanums", "signals")
feats = synth.synthetic_features(B, seed=78)
w = W.random_weights(seed=5, lstm_bias_std=0.1)
ref = Engine(max_batch=B, precision=prec, slots=1)
ref.load_weights(w)
ref_act, ref_pred = ref.run(*(feats[k] for k in keys))
ref.close()
eng = Engine(max_batch=B, precision=prec, slots=slots, shared_event_stream=True)
eng.load_weights(w)
dev = torch.device("cuda", 0)
d = {k: torch.from_numpy(feats[k]).to(dev) for k in keys}
ra, rp = torch.from_numpy(ref_act).to(dev), torch.from_numpy(ref_pred).to(dev)
NB = 4 * slots                      # forwards per round: every slot several times, then one sync and NB comparisons on the device
oa = torch.empty((NB, B, 2), dtype=torch.float32, device=dev)
op = torch.empty((NB, B), dtype=torch.int32, device=dev)
t0 = time.time(); forwards = 0; bad = 0; bad_tiles = 0
while forwards * B < min_sites and time.time() - t0 < max_s:
    for i in range(NB):
        eng.run_device(B, *(d[k].data_ptr() for k in keys), oa[i].data_ptr(), op[i].data_ptr())
    eng.sync()
    neq = (oa.view(torch.int32) != ra.view(torch.int32)[None]).any(dim=2) | (op != rp[None])      # [NB, B] sites that differ
    nb = int(neq.any(dim=1).sum())
    if nb:
        bad += nb
        bad_tiles += int(neq.view(NB, -1, 32).any(dim=2).sum()) if B % 32 == 0 else 0
    forwards += NB
eng.close()
rec = {"precision": prec, "batch": B, "slots": slots, "configuration": "eager issue, every slot's BiLSTM on one shared stream, identical data per slot",
       "forwards": forwards, "sites": forwards * B, "seconds": round(time.time() - t0, 1), "mismatching_forwards": bad, "mismatching_32_site_tiles": bad_tiles}
print(json.dumps(rec))
sys.exit(1 if bad else 0)
