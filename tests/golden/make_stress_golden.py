"""Generate tests/golden/stress_golden.npz -- the trained-regime weight set of the parity tests (VERDICT r03 "weak" #1).

`weights.random_weights` (TF-initializer style) keeps the whole network in its linear regime: LSTM |h| <= 0.8, logits within
+-0.7 and ONE label for every site of a batch, so label checks on it assert nothing. This script builds the two balanced
heads the tests use and pins a small batch of float64-oracle outputs for the stress set:

  * `stress_head`: `dense_1/kernel` for `weights.stress_weights(STRESS_SEED)` -- logits of standard deviation 3.5 (span
    about +-10), anti-correlated columns, labels balanced (`weights.centred_head` on the float64 oracle's fc1 of a probe batch);
  * `small_head`: the same construction at logit std 0.5 for the benign `random_weights(seed=7, lstm_bias_std=0.1)` set the
    `small_weights` fixture has always used (so that every old `pred == o_pred` check sees both labels);
  * inputs of 48 sites (all-N k-mer, truncated and all-zero signal windows among them) with the float64 oracle's act /
    pred / logits / top-layer LSTM outputs on the stress set, written only if the independent PyTorch float64 statement
    (oracle/torch_statement.py) agrees.

tests/golden/make_tf_golden.py replays the same inputs and weights through TensorFlow (the stress regime is then part of
the TensorFlow pin). Run: python tests/golden/make_stress_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from deepsignal_amd import synth, weights  # noqa: E402
from oracle import oracle  # noqa: E402
from oracle import torch_statement  # noqa: E402

SMALL_SEED, SMALL_BIAS_STD, SMALL_LOGIT_STD = 7, 0.1, 0.5
STRESS_LOGIT_STD = 3.5
PROBE_N, N = 96, 48


def head_for(w, seed, logit_std):
    probe = synth.synthetic_features(PROBE_N, seed=seed + 2)
    _, _, taps = oracle.forward(w, probe, "f64", taps=True)
    return weights.centred_head(taps["fc1"], w["dense_1/kernel"][:, 0], logit_std, seed + 3)


small = weights.random_weights(seed=SMALL_SEED, lstm_bias_std=SMALL_BIAS_STD)
small_head = head_for(small, SMALL_SEED, SMALL_LOGIT_STD)
stress = weights.stress_weights(weights.STRESS_SEED)
stress_head = head_for(stress, weights.STRESS_SEED, STRESS_LOGIT_STD)
weights.install_head(stress, stress_head)

f = synth.synthetic_features(N, seed=4242)
f["kmer"][0, :] = 4
f["signals"][1, 40:] = 0.0
f["signals"][2, :] = 0.0
f["sanums"][3, :] = 200.0
act, pred, taps = oracle.forward(stress, f, "f64", taps=True)
t_act, t_pred, t_taps = torch_statement.forward(stress, f, torch.float64, True)
assert np.abs(act - t_act).max() < 1e-9 and (pred == t_pred).all()
for k in taps:
    assert np.abs(taps[k] - t_taps[k]).max() < 1e-6 * max(1.0, np.abs(taps[k]).max()), k
assert 0.25 < pred.mean() < 0.75, pred.mean()
out = {"stress_seed": weights.STRESS_SEED, "stress_head": stress_head, "stress_logit_std": STRESS_LOGIT_STD,
       "small_seed": SMALL_SEED, "small_lstm_bias_std": SMALL_BIAS_STD, "small_head": small_head,
       "small_logit_std": SMALL_LOGIT_STD,
       "act": act, "pred": pred, "logits": taps["logits"],
       "lstm_fw_l2_last": taps["lstm_fw_l2"][:, -1, :], "lstm_bw_l2_first": taps["lstm_bw_l2"][:, 0, :],
       "module11_site0": taps["module11"][0]}
out.update({"in_" + k: v for k, v in f.items()})
np.savez_compressed(os.path.join(HERE, "stress_golden.npz"), **out)
print("wrote stress_golden.npz: logits in [%.2f, %.2f], label-1 share %.2f, max |h| %.4f" % (
    taps["logits"].min(), taps["logits"].max(), pred.mean(), np.abs(taps["lstm_fw_l2"]).max()))
