"""Generate tests/golden/forward_golden.npz — golden vectors of the forward pass.

The reference's forward arithmetic (TensorFlow 1.x) cannot run anywhere here (SURVEY.md F5/F7:
parity unpinned by the reference), so the vectors come from the build's own float64 oracle and are
only written if the independent PyTorch float64 statement agrees to < 1e-9 on every tensor.
Run: python tests/golden/make_forward_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from deepsignal_amd import synth, weights  # noqa: E402
from oracle import oracle  # noqa: E402
from oracle import torch_statement  # noqa: E402

WEIGHT_SEED, FEATURE_SEED, N = 7, 2024, 12
w = weights.random_weights(seed=WEIGHT_SEED, lstm_bias_std=0.1)
f = synth.synthetic_features(N, seed=FEATURE_SEED)
f["kmer"][0, :] = 4
f["signals"][1, 40:] = 0.0
act, pred, taps = oracle.forward(w, f, "f64", taps=True)
t_act, t_pred, t_taps = torch_statement.forward(w, f, torch.float64, True)
assert np.abs(act - t_act).max() < 1e-9 and (pred == t_pred).all()
for k in taps:
    assert np.abs(taps[k] - t_taps[k]).max() < 1e-6 * max(1.0, np.abs(taps[k]).max()), k
out = {"weight_seed": WEIGHT_SEED, "lstm_bias_std": 0.1, "act": act, "pred": pred, "logits": taps["logits"],
       "lstm_fw_l2_last": taps["lstm_fw_l2"][:, -1, :], "lstm_bw_l2_first": taps["lstm_bw_l2"][:, 0, :],
       "stem_pool_site0": taps["stem_pool"][0], "module1_site0": taps["module1"][0], "module4_site1": taps["module4"][1],
       "module11": taps["module11"], "signal_feat_head": taps["signal_feat"][:, :512], "fc1_head": taps["fc1"][:, :512]}
out.update({"in_" + k: v for k, v in f.items()})
np.savez_compressed(os.path.join(HERE, "forward_golden.npz"), **out)
print("wrote forward_golden.npz", {k: getattr(v, "shape", v) for k, v in out.items()})
