"""Synthetic (kmer=17, signal=360) feature batches — the distributions fixed in SURVEY.md section 8(d).

Shapes/dtypes are those of the reference placeholders (`model.py:31-37`); the value distributions
imitate what `extract_features.py:215-286` produces (normalised signal, 6-dp rounding, zero-padded
tails for short windows `extract_features.py:157-160`, CpG at the k-mer centre).
"""
from __future__ import annotations

from typing import Dict

import numpy as np

FEATURE_SEED = 17360


def synthetic_features(n: int, seed: int = FEATURE_SEED, kmer_len: int = 17,
                       signal_len: int = 360) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    kmer = rng.integers(0, 4, size=(n, kmer_len), dtype=np.int32)
    c = (kmer_len - 1) // 2
    n_mask = rng.random((n, kmer_len)) < 0.001
    kmer[n_mask] = 4
    kmer[:, c] = 1                       # 'C'
    if c + 1 < kmer_len:
        kmer[:, c + 1] = 2               # 'G'
    means = np.round(np.clip(rng.normal(0.0, 1.0, (n, kmer_len)), -5, 5), 6).astype(np.float32)
    stds = np.round(np.abs(rng.normal(0.15, 0.08, (n, kmer_len))) + 0.01, 6).astype(np.float32)
    sanums = np.minimum(1 + rng.poisson(8, (n, kmer_len)), 200).astype(np.float32)
    signals = np.round(np.clip(rng.normal(0.0, 1.0, (n, signal_len)), -5, 5), 6).astype(np.float32)
    short = np.nonzero(rng.random(n) < 0.05)[0]
    for i in short:
        keep = int(rng.integers(kmer_len, signal_len))
        signals[i, keep:] = 0.0
    labels = rng.integers(0, 2, size=n, dtype=np.int32)
    return {"kmer": kmer, "means": means, "stds": stds, "sanums": sanums,
            "signals": signals, "labels": labels}
