"""Locate the TensorFlow-written fixtures of tests/golden/make_tf_golden.py. They can only be produced where TensorFlow
1.8 - 1.13 and the reference run (not in the build container, not on the GPU box), so every consumer SKIPS LOUDLY while
they are absent -- and pins SURVEY.md row 8c (oracle vs TensorFlow) and row f3 (checkpoint importer vs a file TensorFlow
wrote) the day someone runs that one script."""
import os
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TF_DIR = os.environ.get("DS_TF_GOLDEN_DIR", os.path.join(ROOT, "tests", "golden", "tf"))
HOWTO = ("PARITY STAYS UNPINNED AGAINST TENSORFLOW: %s is absent. Under TensorFlow 1.8 - 1.13 run "
         "`python tests/golden/make_tf_golden.py --reference <deepsignal checkout>` and commit tests/golden/tf/tf_golden.npz "
         "(keep model.ckpt.* next to it or set DS_TF_GOLDEN_DIR).")


def golden():
    path = os.path.join(TF_DIR, "tf_golden.npz")
    if not os.path.exists(path):
        pytest.skip(HOWTO % path)
    return np.load(path)


def checkpoint_prefix():
    prefix = os.path.join(TF_DIR, "model.ckpt")
    if not (os.path.exists(prefix + ".index") and os.path.exists(prefix + ".data-00000-of-00001")):
        pytest.skip(HOWTO % (prefix + ".{index,data-00000-of-00001}"))
    return prefix


def weights_of(g):
    """The weights make_tf_golden.py assigned: regenerated from the recorded seed, every tensor checked against the
    recorded CRC-32 (a numpy whose Generator stream differs would otherwise compare two different models)."""
    from deepsignal_amd import weights as W
    w = W.random_weights(seed=int(g["weight_seed"]), lstm_bias_std=float(g["lstm_bias_std"]))
    for name, crc in zip(g["weight_names"], g["weight_crc32"]):
        got = zlib.crc32(np.ascontiguousarray(w[str(name)], dtype="<f4").tobytes())
        assert got == int(crc), "weight tensor %s regenerated from the seed differs from the one TensorFlow was given" % name
    return w


def features_of(g):
    return {k: g["in_" + k] for k in ("kmer", "means", "stds", "sanums", "signals")}
