#!/usr/bin/env python
"""Generate the TensorFlow-tied pin of the forward arithmetic ("parity unpinned").

CANNOT RUN IN THE BUILD CONTAINER OR ON THE GPU BOX: it needs TensorFlow 1.8 - 1.13 (README.md:28 of the reference; needs
`tf.contrib`) and the reference checkout. Anyone who has such an environment runs it ONCE (tests/golden/README.md has the
Docker one-liner and the expected output):

    python tests/golden/make_tf_golden.py --reference /path/to/deepsignal [--out tests/golden/tf]

What it does (the reference's own code path, nothing of this repository's arithmetic):
  1. builds the reference graph `Model(base_num=17, signal_num=360, class_num=2)`   (deepsignal/model.py:25-108, as
     call_modifications.py:203-206 does);
  2. assigns this repository's seeded random weights (deepsignal_amd.weights.random_weights, pure numpy; or
     --weights FILE.dsw, a DSAMDW01 file) to the graph variables BY THE NAMES of SURVEY.md Appendix B.7 -- a variable of
     spec.tensor_table the graph does not have, or a trainable graph variable the table does not name, aborts: that is
     the first-contact check of the name map;
  3. saves a V2 checkpoint with tf.train.Saver() exactly as train_model.py:33-36,242-243 does (Adam slots,
     global_step and the BN zero-debias variables included) -> <out>/model.ckpt.{index,data-00000-of-00001};
  4. RESTORES it into a fresh session (call_modifications.py:208-212) and runs
     sess.run([model.activation_logits, model.prediction], feed_dict) with the feed of call_modifications.py:168-178
     (training False, keep_prob 1.0) on the inputs of tests/golden/forward_golden.npz, in one batch and in batches of 5;
  5. runs the same graph on the trained-regime STRESS set (weights.stress_weights + the committed head and inputs of
     tests/golden/stress_golden.npz) and records TensorFlow's outputs there too;
  6. writes <out>/tf_golden.npz: inputs, TensorFlow's activation_logits / prediction, the weight seed and a CRC-32 of
     every weight tensor as assigned (so a consumer regenerating the weights from the seed can tell whether it got the
     same numbers), TensorFlow's version string.

Consumers (skip LOUDLY while the fixture is absent): tests/test_tf_golden.py (CPU: oracle vs TensorFlow, checkpoint
importer vs the assigned tensors) and tests/test_gpu_configs.py::test_tf_written_checkpoint_on_the_gpu (HIP engine fed the
imported checkpoint vs TensorFlow's outputs, 1e-4). Commit tf_golden.npz (a few kB); the checkpoint pair is ~485 MB --
keep it next to the npz (the tests look there) or point DS_TF_GOLDEN_DIR at the directory.
"""
import argparse
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--reference", required=True, help="checkout of bioinfomaticsCSU/deepsignal (the directory holding deepsignal/)")
    ap.add_argument("--out", default=os.path.join(HERE, "tf"))
    ap.add_argument("--weights", default=None, help="DSAMDW01 weight file to assign instead of regenerating from the seed")
    args = ap.parse_args()

    import tensorflow as tf
    major, minor = (int(x) for x in tf.__version__.split(".")[:2])
    if (major, minor) < (1, 8) or (major, minor) > (1, 15):
        raise SystemExit("TensorFlow %s: the reference needs 1.8.0 <= v <= 1.13.1 (tf.contrib)" % tf.__version__)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.abspath(args.reference))
    from deepsignal_amd import spec, weights as W        # pure numpy
    from deepsignal.model import Model                   # the reference graph

    g = np.load(os.path.join(HERE, "forward_golden.npz"))
    seed, bias_std = int(g["weight_seed"]), float(g["lstm_bias_std"])
    w = W.load_weights(args.weights) if args.weights else W.random_weights(seed=seed, lstm_bias_std=bias_std)
    W.check_weights(w)
    feats = {k: g["in_" + k] for k in ("kmer", "means", "stds", "sanums", "signals", "labels")}
    n = feats["kmer"].shape[0]
    os.makedirs(args.out, exist_ok=True)
    prefix = os.path.join(args.out, "model.ckpt")

    def feed(model, s, e):       # call_modifications.py:168-176 (lists of Python numbers there; arrays feed alike)
        return {model.base_int: feats["kmer"][s:e], model.means: feats["means"][s:e], model.stds: feats["stds"][s:e],
                model.sanums: feats["sanums"][s:e], model.signals: feats["signals"][s:e], model.labels: feats["labels"][s:e],
                model.lr: 0.001, model.training: False, model.keep_prob: 1.0}

    # ---- build, assign, save
    tf.reset_default_graph()
    model = Model(base_num=17, signal_num=360, class_num=2)
    gvars = {v.op.name: v for v in tf.global_variables()}
    missing = [name for name, _ in spec.tensor_table() if name not in gvars]
    if missing:
        raise SystemExit("graph lacks variables of spec.tensor_table (SURVEY.md Appendix B.7 name map is wrong): %s" % missing[:8])
    named = set(name for name, _ in spec.tensor_table())
    unnamed = [v.op.name for v in tf.trainable_variables() if v.op.name not in named]
    if unnamed:
        raise SystemExit("trainable graph variables the name map does not cover: %s" % unnamed[:8])
    with tf.Session() as sess:
        sess.run(tf.global_variables_initializer())
        for name, shape in spec.tensor_table():
            v = gvars[name]
            if tuple(v.shape.as_list()) != tuple(shape):
                raise SystemExit("%s: graph shape %s, table shape %s" % (name, v.shape.as_list(), shape))
            v.load(w[name], sess)
        tf.train.Saver().save(sess, prefix, write_meta_graph=False)
        act_direct, pred_direct = sess.run([model.activation_logits, model.prediction], feed_dict=feed(model, 0, n))

    # ---- restore as call_mods does, run
    tf.reset_default_graph()
    model = Model(base_num=17, signal_num=360, class_num=2)
    with tf.Session() as sess:
        tf.train.Saver().restore(sess, prefix)
        sess.run(tf.local_variables_initializer())
        act, pred = sess.run([model.activation_logits, model.prediction], feed_dict=feed(model, 0, n))
        parts = [sess.run([model.activation_logits, model.prediction], feed_dict=feed(model, s, min(n, s + 5))) for s in range(0, n, 5)]
    act5 = np.concatenate([p[0] for p in parts])
    pred5 = np.concatenate([p[1] for p in parts])
    assert np.array_equal(act, act_direct) and np.array_equal(pred, pred_direct), "restore changed the outputs"

    # ---- the trained-regime stress set (tests/golden/stress_golden.npz: saturating LSTM gates, hot BN channels, logits spanning
    # +-10, both labels): same graph, weights assigned in place, no checkpoint -- so the TensorFlow pin covers the regime the
    # benign random-init weights never visit
    sg = np.load(os.path.join(HERE, "stress_golden.npz"))
    ws = W.stress_weights(int(sg["stress_seed"]), head=sg["stress_head"])
    W.check_weights(ws)
    sfeats = {k: sg["in_" + k] for k in ("kmer", "means", "stds", "sanums", "signals", "labels")}
    tf.reset_default_graph()
    model = Model(base_num=17, signal_num=360, class_num=2)
    gvars = {v.op.name: v for v in tf.global_variables()}
    with tf.Session() as sess:
        sess.run(tf.global_variables_initializer())
        for name, _ in spec.tensor_table():
            gvars[name].load(ws[name], sess)
        ns = sfeats["kmer"].shape[0]
        s_act, s_pred = sess.run([model.activation_logits, model.prediction], feed_dict={
            model.base_int: sfeats["kmer"], model.means: sfeats["means"], model.stds: sfeats["stds"], model.sanums: sfeats["sanums"],
            model.signals: sfeats["signals"], model.labels: sfeats["labels"], model.lr: 0.001, model.training: False,
            model.keep_prob: 1.0})
    print("stress set: %d sites, label-1 share %.2f, max |act - float64 oracle (committed)| = %.3g"
          % (ns, float(np.mean(s_pred)), float(np.abs(s_act - sg["act"]).max())))

    out = {"tf_version": tf.__version__, "weight_seed": seed, "lstm_bias_std": bias_std,
           "stress_act": s_act.astype(np.float32), "stress_pred": s_pred.astype(np.int64), "stress_seed": int(sg["stress_seed"]),
           "act": act.astype(np.float32), "pred": pred.astype(np.int64), "act_batches_of_5": act5.astype(np.float32),
           "pred_batches_of_5": pred5.astype(np.int64),
           "weight_names": np.array([name for name, _ in spec.tensor_table()]),
           "weight_crc32": np.array([zlib.crc32(np.ascontiguousarray(w[name], dtype="<f4").tobytes()) for name, _ in spec.tensor_table()],
                                    dtype=np.uint32),
           "checkpoint_variables": np.array(sorted(gvars))}
    out.update({"in_" + k: v for k, v in feats.items()})
    np.savez_compressed(os.path.join(args.out, "tf_golden.npz"), **out)
    print("TensorFlow %s: wrote %s/tf_golden.npz and %s.{index,data-00000-of-00001}" % (tf.__version__, args.out, prefix))
    print("max |act(one batch) - act(batches of 5)| =", float(np.abs(act - act5).max()))


if __name__ == "__main__":
    main()
