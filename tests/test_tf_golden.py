"""CPU: the oracle and the checkpoint importer against what TensorFlow itself produced (tests/golden/make_tf_golden.py).
Skipped, loudly, until that script has been run somewhere TensorFlow 1.x exists -- see tests/tf_golden_fixture.py."""
import numpy as np

import tf_golden_fixture as fx


def test_generator_script_is_self_contained():
    """The generator must stay runnable by someone with nothing but TensorFlow 1.x, numpy and the two checkouts: it may
    import only the pure-numpy parts of this package (never the oracle, the engine or torch)."""
    import ast
    import os
    src = open(os.path.join(fx.ROOT, "tests", "golden", "make_tf_golden.py")).read()
    mods = set()
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.Import):
            mods.update(a.name.split(".")[0] for a in node.names)
        elif isinstance(node, ast.ImportFrom) and node.module:
            mods.add(node.module.split(".")[0] if node.module.split(".")[0] != "deepsignal_amd" else node.module)
    assert mods <= {"argparse", "os", "sys", "zlib", "numpy", "tensorflow", "deepsignal_amd", "deepsignal"}, mods
    # and the consumers find what it writes
    assert "tf_golden.npz" in src and "model.ckpt" in src


def test_oracle_matches_tensorflow():
    """Row 8c: the fp32 oracle within 1e-5 of TensorFlow's sigmoid outputs (the bar the HIP path is held to against the
    oracle), the float64 oracle within 2e-6, labels equal wherever the margin exceeds 1e-3."""
    from oracle import oracle
    g = fx.golden()
    w, feats = fx.weights_of(g), fx.features_of(g)
    act32, pred32 = oracle.forward(w, feats, "f32")
    act64, _ = oracle.forward(w, feats, "f64")
    tf_act, tf_pred = g["act"], g["pred"]
    assert np.abs(act32 - tf_act).max() <= 1e-5, "fp32 oracle vs TensorFlow %s: %g" % (g["tf_version"], np.abs(act32 - tf_act).max())
    assert np.abs(act64 - tf_act).max() <= 2e-6
    decided = np.abs(tf_act[:, 1] - tf_act[:, 0]) > 1e-3
    assert (pred32[decided] == tf_pred[decided]).all()
    # TensorFlow's own batching invariance, for the record (a site's result must not depend on its batch mates)
    assert np.abs(g["act"] - g["act_batches_of_5"]).max() <= 1e-6


def test_checkpoint_importer_reads_the_file_tensorflow_wrote():
    """Row f3: every model tensor comes out of the TensorFlow-written checkpoint bit for bit, optimizer slots,
    global_step and the zero-debias variables are ignored, every block / tensor checksum verifies."""
    from deepsignal_amd import tf_checkpoint
    g = fx.golden()
    prefix = fx.checkpoint_prefix()
    w = fx.weights_of(g)
    got = tf_checkpoint.checkpoint_to_weights(prefix)
    assert set(got) == set(w)
    for name in w:
        assert got[name].shape == w[name].shape and np.array_equal(got[name], w[name]), name
    _, entries = tf_checkpoint.read_index(prefix)
    assert set(str(v) for v in g["checkpoint_variables"]) <= set(entries), "variables TensorFlow saved that the index reader did not list"


def test_oracle_matches_tensorflow_on_the_stress_set():
    """Row 8c in the regime a trained model lives in (saturating LSTM gates, logits spanning +-10, both labels): TensorFlow's
    fp32 outputs within 1e-4 of the float64 oracle (fp32 evaluations of this set differ from float64 by 3 - 5e-5 whatever
    the summation order: tests/test_gpu_stress.py), labels equal wherever the margin exceeds 1e-3, both labels present."""
    import os
    from oracle import oracle
    from deepsignal_amd import weights as W
    g = fx.golden()
    if "stress_act" not in g.files:
        import pytest
        pytest.skip("tf_golden.npz predates the stress set: re-run tests/golden/make_tf_golden.py")
    sg = np.load(os.path.join(fx.ROOT, "tests", "golden", "stress_golden.npz"))
    assert int(g["stress_seed"]) == int(sg["stress_seed"])
    w = W.stress_weights(int(sg["stress_seed"]), head=sg["stress_head"])
    feats = {k: sg["in_" + k] for k in ("kmer", "means", "stds", "sanums", "signals")}
    a64, p64 = oracle.forward(w, feats, "f64")
    tf_act, tf_pred = g["stress_act"], g["stress_pred"]
    assert np.abs(a64 - tf_act).max() <= 1e-4
    decided = np.abs(a64[:, 1] - a64[:, 0]) > 1e-3
    assert (p64[decided] == tf_pred[decided]).all()
    assert 0.2 < float(np.mean(tf_pred)) < 0.8
