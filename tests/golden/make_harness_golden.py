#!/opt/conda/bin/python3.9
"""Generate harness golden vectors by RUNNING THE REFERENCE's own Python harness.

Run here (build container only):
    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/make_harness_golden.py

TensorFlow 1.x cannot be installed, so `tensorflow*` is stubbed in sys.modules and
the reference's `_call_mods` / `_read_features_file` are driven with a fake session that returns
canned activations. What this pins (and what the committed JSON holds — data only, no reference
source): (i) the slicing of a queue item into batch_size chunks and the feed contents per chunk,
(ii) float32 prob normalisation p/(p0+p1) and the exact output row text (str(np.float32)),
(iii) TSV parsing and the read-grouping of `_read_features_file` into queue items.
Reference: /root/reference/deepsignal/call_modifications.py:35-91,149-194.
"""
import json
import os
import sys
import tempfile
from unittest import mock

import numpy as np

np.int = int        # removed aliases the reference still uses
np.float = float
for name in ("tensorflow", "tensorflow.contrib", "tensorflow.contrib.rnn", "tensorflow.contrib.layers",
             "tensorflow.contrib.framework"):
    sys.modules[name] = mock.MagicMock()
sys.path.insert(0, "/root/reference")
from deepsignal import call_modifications as ref   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(20190417)

KMER, SIG = 17, 360
BASES = "ACGTN"


def make_rows(nreads, sites_per_read):
    rows = []
    for r in range(nreads):
        for s in range(sites_per_read[r]):
            kmer = "".join(BASES[i] for i in rng.integers(0, 4, KMER))
            kmer = kmer[:8] + "CG" + kmer[10:]
            if rng.random() < 0.2:
                kmer = "N" + kmer[1:]
            means = np.round(rng.normal(0, 1, KMER), 6)
            stds = np.round(np.abs(rng.normal(0.15, 0.08, KMER)) + 0.01, 6)
            lens = 1 + rng.poisson(8, KMER)
            sig = np.round(np.clip(rng.normal(0, 1, SIG), -5, 5), 6)
            if s == 0:
                sig[200:] = 0.0
            rows.append("\t".join([
                "chr%d" % (1 + r % 3), str(1000 * r + 7 * s), "+-"[s % 2], str(5000 - s), "read_%03d" % r, "tc"[r % 2],
                kmer, ",".join(str(x) for x in means), ",".join(str(x) for x in stds),
                ",".join(str(int(x)) for x in lens), ",".join(str(x) for x in sig), str(int(rng.integers(0, 2)))]))
    return rows


class FakeQueue(list):
    def put(self, x):
        self.append(x)

    def qsize(self):
        return 0


class FakeModel:
    base_int, means, stds, sanums, signals, labels, lr, training, keep_prob = range(9)
    activation_logits, prediction = "act", "pred"


class FakeSession:
    """Returns canned sigmoid outputs; records what the reference fed."""

    def __init__(self):
        self.calls = []

    def run(self, fetches, feed_dict):
        n = len(feed_dict[FakeModel.base_int])
        act = rng.uniform(0.02, 0.98, size=(n, 2)).astype(np.float32)
        if n > 2:
            act[1] = act[1, 0]                 # exact tie -> argmax picks label 0
            act[2] = (np.float32(1.0), np.float32(1e-7))
        pred = np.argmax(act, axis=1)
        self.calls.append({
            "n": n,
            "kmer_first": [int(x) for x in feed_dict[FakeModel.base_int][0]],
            "sanums_first": [float(x) for x in feed_dict[FakeModel.sanums][0]],
            "means_first": [float(x) for x in feed_dict[FakeModel.means][0]],
            "signals_first_head": [float(x) for x in feed_dict[FakeModel.signals][0][:5]],
            "training": bool(feed_dict[FakeModel.training]), "keep_prob": float(feed_dict[FakeModel.keep_prob]),
            "act": [[float(a), float(b)] for a, b in act],
        })
        return act, pred


def main():
    cases = []
    for name, nreads, spr, f5_batch_num, batch_size in [
            ("two_reads_per_item", 5, [3, 1, 4, 2, 5], 2, 4),
            ("one_item_ragged_tail", 3, [7, 6, 4], 50, 5),
            ("single_row", 1, [1], 1, 512)]:
        rows = make_rows(nreads, spr)
        with tempfile.NamedTemporaryFile("w", suffix=".tsv", delete=False) as f:
            f.write("\n".join(rows) + "\n")
            path = f.name
        q = FakeQueue()
        ref._read_features_file(path, q, f5_batch_num)
        os.unlink(path)
        assert q[-1] == "kill"
        items = q[:-1]
        sess = FakeSession()
        outs = []
        for item in items:
            pred_str, accuracy, batch_num = ref._call_mods(item, sess, FakeModel, 0.001, batch_size)
            outs.append({"pred_str": pred_str, "accuracy": float(accuracy), "batch_num": int(batch_num)})
        cases.append({
            "name": name, "f5_batch_num": f5_batch_num, "batch_size": batch_size, "tsv_rows": rows,
            "queue_items": [{"n": len(it[0]), "sampleinfo": it[0], "kmers": it[1], "labels": [int(x) for x in it[6]],
                             "lens_first": [int(x) for x in it[4][0]]} for it in items],
            "session_calls": sess.calls, "outputs": outs,
        })
    with open(os.path.join(HERE, "harness_golden.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_harness_golden.py (reference harness under stub-TF)", "cases": cases}, f)
    print("wrote harness_golden.json:", [(c["name"], len(c["queue_items"]), len(c["session_calls"])) for c in cases])


if __name__ == "__main__":
    main()
