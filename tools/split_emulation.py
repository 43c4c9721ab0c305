"""CPU emulation of the split-operand engine (stage 0): how far from the float64 oracle does
fp32-accumulated bf16-term arithmetic land on the three weight sets of the parity tests?

    python tools/split_emulation.py [n_sites] [out.json]

Rows: the native fp32 oracle (what the fp32 engine is held to), oracle/torch_statement.forward_split with three terms /
six products and two terms / three products, split in the fused inception chain only (stage 1 of the plan) and in every
matrix product of the path (modules + conv_layer2/3 + BiLSTM + dense 6032 x 6032), and forward_bf16 (the shipped bf16 modes'
rounding points). Columns per weight set: mean / max abs difference of the sigmoid outputs to the float64 oracle, and the share
of sites whose label differs from float64's. No GPU involved."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepsignal_amd import synth, weights  # noqa: E402
from oracle import oracle, torch_statement as ts  # noqa: E402


def weight_sets():
    g = np.load(os.path.join(ROOT, "tests", "golden", "stress_golden.npz"))
    small = weights.random_weights(seed=7, lstm_bias_std=0.1)
    bal = dict(small)
    weights.install_head(bal, g["small_head"])
    return {"benign": weights.random_weights(), "balanced": bal,
            "stress": weights.stress_weights(int(g["stress_seed"]), head=g["stress_head"])}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    out_path = sys.argv[2] if len(sys.argv) > 2 else None
    torch.set_num_threads(oracle.usable_cores())
    everything = ("modules", "stem23", "lstm", "fc1")
    arms = [
        ("native fp32 (oracle/ds_oracle.c, REAL=float)", None),
        ("split3 (6 products), inception chain only", dict(terms=3, scope=("modules",))),
        ("split3 (6 products), every matrix product", dict(terms=3, scope=everything)),
        ("split2 (3 products), inception chain only", dict(terms=2, scope=("modules",))),
        ("split2 (3 products), every matrix product", dict(terms=2, scope=everything)),
        ("split3, every matrix product, float64 accumulate (the dropped products alone)", dict(terms=3, scope=everything, acc_dtype=torch.float64)),
        ("bf16 (shipped rounding points)", "bf16"),
        ("bf16_all (shipped rounding points)", "bf16_all"),
    ]
    table = {}
    for wname, w in weight_sets().items():
        feats = synth.synthetic_features(n, seed=900 + n)
        a64, p64 = oracle.forward(w, feats, "f64")
        for label, kw in arms:
            if kw is None:
                act, pred = oracle.forward(w, feats, "f32")
            elif kw == "bf16":
                act, pred = ts.forward_bf16(w, feats)
            elif kw == "bf16_all":
                act, pred = ts.forward_bf16(w, feats, lstm_bf16=True)
            else:
                act, pred = ts.forward_split(w, feats, **kw)
            d = np.abs(act.astype(np.float64) - a64)
            rec = {"mean_abs_d_act": float(d.mean()), "max_abs_d_act": float(d.max()),
                   "label_flips": float((pred != p64).mean()), "label1_share_f64": float(p64.mean())}
            table.setdefault(label, {})[wname] = rec
            print("%-66s %-9s mean %.2e  max %.2e  flips %.3f" % (label, wname, rec["mean_abs_d_act"], rec["max_abs_d_act"], rec["label_flips"]), flush=True)
    if out_path:
        json.dump({"n_sites": n, "arbiter": "oracle/ds_oracle.c REAL=double", "table": table}, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
