"""CPU: the from-scratch call_mods harness against golden vectors captured from the REFERENCE's
own harness (tests/golden/make_harness_golden.py ran deepsignal/call_modifications.py under
stub-TensorFlow): TSV parsing, read grouping, batch slicing, feed contents, float32 probability
normalisation and the exact output row text."""
import json
import os

import numpy as np
import pytest

from deepsignal_amd import call_modifications as cm
from deepsignal_amd.utils.process_utils import base2code_dna, code2base_dna, str2bool

GOLD = os.path.join(os.path.dirname(__file__), "golden", "harness_golden.json")


@pytest.fixture(scope="module")
def cases():
    with open(GOLD) as f:
        return json.load(f)["cases"]


class ReplayEngine:
    """Returns the activations the reference's fake session returned, and records the feeds."""

    def __init__(self, calls):
        self.calls = list(calls)
        self.i = 0
        self.seen = []

    def run(self, kmer, means, stds, sanums, signals):
        c = self.calls[self.i]
        self.i += 1
        assert len(kmer) == c["n"]
        self.seen.append({"kmer_first": [int(x) for x in kmer[0]], "sanums_first": [float(x) for x in sanums[0]],
                          "means_first": [float(x) for x in means[0]],
                          "signals_first_head": [float(x) for x in signals[0][:5]]})
        act = np.asarray(c["act"], dtype=np.float32)
        return act, np.argmax(act, axis=1)


class ListQueue(list):
    def put(self, x):
        self.append(x)


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_reader_and_call_mods_match_reference(cases, idx, tmp_path):
    case = cases[idx]
    path = str(tmp_path / "features.tsv")
    with open(path, "w") as f:
        f.write("\n".join(case["tsv_rows"]) + "\n")
    q = ListQueue()
    cm._read_features_file(path, q, case["f5_batch_num"])
    assert q[-1] == "kill"
    items = q[:-1]
    assert len(items) == len(case["queue_items"])
    for it, ref in zip(items, case["queue_items"]):
        assert len(it[0]) == ref["n"] and it[0] == ref["sampleinfo"] and it[1] == ref["kmers"]
        assert [int(x) for x in it[6]] == ref["labels"] and [int(x) for x in it[4][0]] == ref["lens_first"]
    eng = ReplayEngine(case["session_calls"])
    for it, ref in zip(items, case["outputs"]):
        pred_str, accuracy, batch_num = cm._call_mods(it, eng, case["batch_size"])
        assert pred_str == ref["pred_str"]            # exact text, incl. str(np.float32) and tie -> label 0
        assert batch_num == ref["batch_num"] and abs(accuracy - ref["accuracy"]) < 1e-12
    assert eng.i == len(case["session_calls"])
    for seen, ref in zip(eng.seen, case["session_calls"]):
        for k in seen:
            assert seen[k] == ref[k], k


def test_call_mods_end_to_end_file(cases, tmp_path):
    """Feature file -> result file through call_mods(), order and grouping preserved."""
    case = cases[0]
    path, out = str(tmp_path / "f.tsv"), str(tmp_path / "r.tsv")
    with open(path, "w") as f:
        f.write("\n".join(case["tsv_rows"]) + "\n")
    eng = ReplayEngine(case["session_calls"])
    n = cm.call_mods(path, "unused", out, 17, 360, case["batch_size"], 0.001, 2, 1, False, True, True, True,
                     None, engine=eng, f5_batch_num=case["f5_batch_num"])
    expect = [r for o in case["outputs"] for r in o["pred_str"]]
    assert n == len(expect)
    assert open(out).read().splitlines() == expect


def test_writer_and_sentinel(tmp_path):
    import queue
    q = queue.Queue()
    q.put(["a\t1", "b\t2"])
    q.put(["c\t3"])
    q.put("kill")
    out = str(tmp_path / "w.tsv")
    cm._write_predstr_to_file(out, q)
    assert open(out).read() == "a\t1\nb\t2\nc\t3\n"


def test_alphabet_and_flags():
    assert base2code_dna == {"A": 0, "C": 1, "G": 2, "T": 3, "N": 4}
    assert "".join(code2base_dna[i] for i in range(5)) == "ACGTN"
    assert str2bool("yes") and str2bool("True") and str2bool("1") and not str2bool("no")


def test_cli_flag_surface():
    """Same flags / defaults as reference deepsignal.py:236-326."""
    from deepsignal_amd.deepsignal import build_parser
    a = build_parser().parse_args(["call_mods", "-i", "x", "-m", "w", "-o", "o"])
    assert (a.f5_batch_num, a.kmer_len, a.cent_signals_len, a.batch_size, a.class_num) == (50, 17, 360, 512, 2)
    assert (a.is_cnn, a.is_rnn, a.is_base, a.is_gpu, a.nproc) == ("yes", "yes", "yes", "no", 1)
    assert (a.corrected_group, a.basecall_subgroup, a.normalize_method, a.motifs, a.mod_loc) == \
        ("RawGenomeCorrected_000", "BaseCalled_template", "mad", "CG", 0)
    assert a.learning_rate == 0.001 and a.positions is None and a.reference_path is None


def test_fast5_directory_mode(tmp_path, monkeypatch):
    """call_mods on a fast5 directory (config 5 plumbing): files -> host extraction -> engine -> rows.
    The HDF5 access is replaced by the committed raw arrays of the reference-run golden (h5py is not a
    dependency of the test environment); the engine is a deterministic stand-in."""
    from deepsignal_amd import extract_features as ef
    with open(os.path.join(os.path.dirname(GOLD), "extract_golden.json")) as f:
        g = json.load(f)
    d = tmp_path / "f5"
    d.mkdir()
    for name in g["read_order"]:
        (d / (name + ".fast5")).write_bytes(b"")
    (d / "broken.fast5").write_bytes(b"")

    def fake_read(path, corrected_group, basecall_subgroup):
        r = g["reads"][os.path.basename(path)[:-6]]           # KeyError for broken.fast5 -> counted as failed
        return (np.asarray(r["signal"], np.int16), r["starts"], r["lengths"], r["bases"], r["range"] / r["digitisation"],
                r["offset"], (r["read_id"], r["strand"], r["alignstrand"], r["chrom"], r["chrom_start"]))

    monkeypatch.setattr(ef, "_read_fast5", fake_read)

    class Eng:
        def run(self, kmer, means, stds, sanums, signals):
            x = np.asarray(signals, np.float32)
            assert x.shape[1] == 360 and np.asarray(means).shape == np.asarray(kmer).shape
            act = np.stack([1 / (1 + np.exp(-x[:, 0])), 1 / (1 + np.exp(x[:, 1]))], axis=1).astype(np.float32)
            return act, np.argmax(act, axis=1)

    out = str(tmp_path / "r.tsv")
    # the reference's layout (deepsignal.py:83-84): (is_recursive, corrected_group, basecall_subgroup, reference_path,
    # is_dna, normalize_method, motifs, mod_loc, methy_label, f5_batch_num, position_file)
    f5_args = (True, "RawGenomeCorrected_000", "BaseCalled_template", None, True, "mad", "CG", 0, 1, 2, None)
    n = cm.call_mods(str(d), "unused", out, 17, 360, 16, 0.001, 2, 1, False, True, True, True, f5_args, engine=Eng())
    rows = [l.split("\t") for l in open(out).read().splitlines()]
    case = g["cases"][1]                                      # same settings, no reference genome -> pos_in_strand = -1
    assert n == len(rows) == len(case["features_str"])
    expect = sorted("\t".join(r.split("\t")[:7]) for r in case["features_str"])
    assert sorted("\t".join(r[:6] + [r[9]]) for r in rows) == expect


def test_fast5_directory_mode_with_extraction_workers(tmp_path, capsys):
    """nproc > 2: file batches are extracted by spawned workers while this process drives the engine (the reference's
    nproc - 1 extraction processes, call_modifications.py:431-448). Unreadable files are counted, not fatal."""
    d = tmp_path / "f5"
    d.mkdir()
    for i in range(7):
        (d / ("r%d.fast5" % i)).write_bytes(b"")                 # empty files (and no h5py here): every read fails

    class Eng:
        def run(self, *a):
            raise AssertionError("no site should reach the engine")

    out = str(tmp_path / "r.tsv")
    # the reference's layout (deepsignal.py:83-84): (is_recursive, corrected_group, basecall_subgroup, reference_path,
    # is_dna, normalize_method, motifs, mod_loc, methy_label, f5_batch_num, position_file)
    f5_args = (True, "RawGenomeCorrected_000", "BaseCalled_template", None, True, "mad", "CG", 0, 1, 2, None)
    n = cm.call_mods(str(d), "unused", out, 17, 360, 16, 0.001, 2, 4, False, True, True, True, f5_args, engine=Eng())
    assert n == 0 and open(out).read() == ""
    assert "7 of 7 fast5 files failed" in capsys.readouterr().out


def test_f5_args_is_the_references_tuple():
    """call_mods takes f5_args in the reference's order (deepsignal.py:83-84, call_modifications.py:428-429); a tuple in
    any other layout must fail loudly instead of mis-assigning fields."""
    import pytest
    ref = (True, "RawGenomeCorrected_000", "BaseCalled_template", "/ref.fa", True, "zscore", "CG,CHG", 1, 1, 7, "/pos.txt")
    f5 = cm._unpack_f5_args(ref)
    assert (f5.is_recursive, f5.reference_path, f5.normalize_method, f5.motifs, f5.mod_loc, f5.f5_batch_num,
            f5.position_file) == (True, "/ref.fa", "zscore", "CG,CHG", 1, 7, "/pos.txt")
    assert cm._unpack_f5_args(ref, f5_batch_num=3).f5_batch_num == 3
    assert cm._unpack_f5_args(None).f5_batch_num == 50                    # CLI default, deepsignal.py:243
    with pytest.raises(ValueError):
        cm._unpack_f5_args((50,))
    with pytest.raises(ValueError):                                        # round 1's layout: f5_batch_num first
        cm._unpack_f5_args((2, True, "RawGenomeCorrected_000", "BaseCalled_template", True, "mad", "CG", 0, 1, None, None))


def test_cli_builds_the_references_f5_args(monkeypatch):
    from deepsignal_amd import deepsignal as cli
    seen = {}
    monkeypatch.setattr(cm, "call_mods", lambda *a, **k: seen.update(args=a, kw=k))
    cli.main(["call_mods", "-i", "x", "-m", "w", "-o", "o", "--f5_batch_num", "9", "--reference_path", "ref.fa",
              "--positions", "p.txt", "--normalize_method", "zscore"])
    f5 = cm._unpack_f5_args(seen["args"][13])
    assert (f5.f5_batch_num, f5.reference_path, f5.position_file, f5.normalize_method, f5.methy_label) == \
        (9, "ref.fa", "p.txt", "zscore", 1)


class _AsyncStandIn:
    """CPU stand-in with the engine's asynchronous boundary (submit / submit_parts / wait, `slots` in flight): outputs
    are a deterministic function of each site's own inputs, computed at wait() time."""
    class_num, max_batch, slots = 2, 64, 3

    def __init__(self):
        self.pending, self.next, self.parts_calls, self.batches = {}, 0, 0, []

    @staticmethod
    def _f(means, signals):
        p1 = (1.0 / (1.0 + np.exp(-(means.sum(axis=1) + signals[:, :7].sum(axis=1))))).astype(np.float32)
        act = np.stack([1.0 - p1, p1], axis=1).astype(np.float32)
        return act, np.argmax(act, axis=1).astype(np.int32)

    def run(self, kmer, means, stds, sanums, signals):
        return self._f(np.asarray(means, np.float32), np.asarray(signals, np.float32))

    def submit(self, kmer, means, stds, sanums, signals):
        assert len(self.pending) < self.slots, "more batches in flight than slots"
        t = self.next; self.next += 1
        self.pending[t] = (np.array(means, np.float32), np.array(signals, np.float32))
        self.batches.append(len(kmer))
        return (t, len(kmer))

    def submit_parts(self, parts):
        self.parts_calls += 1
        return self.submit(*(np.concatenate([p[j] for p in parts]) for j in range(5)))

    def wait(self, ticket):
        assert ticket[0] == min(self.pending), "tickets must be waited in submission order"
        means, signals = self.pending.pop(ticket[0])
        return self._f(means, signals)


def _items_for_pipeline(n, cuts):
    from deepsignal_amd import fastio, synth
    feats = synth.synthetic_features(n, seed=17)
    items = []
    for s0, e0 in zip(cuts[:-1], cuts[1:]):
        m = e0 - s0
        info = np.frombuffer(("r%04d" * m % tuple(range(s0, e0))).encode(), np.uint8).copy()
        items.append(fastio.FeatureItem(info, np.arange(m + 1, dtype=np.int64) * 5, feats["kmer"][s0:e0], feats["means"][s0:e0],
                                        feats["stds"][s0:e0], feats["sanums"][s0:e0], feats["signals"][s0:e0],
                                        np.zeros(m, np.int32)))
    return feats, items


def test_row_pipeline_threads_keep_file_order_and_fill_batches_across_items():
    """call_mods' row pipeline on a CPU stand-in: batches are filled across queue items (segments handed over with
    submit_parts, no more than `slots` in flight, tickets waited in order), a helper thread formats and sinks the rows,
    and the text is what the blocking per-item path writes."""
    if not os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deepsignal_amd",
                                       "libdeepsignal_hip.so")):
        pytest.skip("native library not built (row formatter)")
    n, cuts = 333, [0, 5, 70, 71, 200, 333]
    feats, items = _items_for_pipeline(n, cuts)
    eng = _AsyncStandIn()
    got = []
    pipe = cm._RowPipeline(eng, 64, lambda tag, data: got.append((tag, data)))
    assert pipe.pipelined
    for tag, it in enumerate(items):
        pipe.feed(it, tag)
        assert pipe.live_tags() <= set(range(tag + 1))
    pipe.flush()
    assert not pipe.live_tags() and not eng.pending
    pipe.close()
    assert eng.batches == [64] * 5 + [13] and eng.parts_calls >= 3          # full batches; straddling ones as segments
    assert [t for t, _ in got] == sorted(t for t, _ in got)
    # the same items through the blocking path (an engine without submit / wait)
    class Blocking:
        class_num = 2
        run = staticmethod(lambda *a: _AsyncStandIn._f(np.asarray(a[1], np.float32), np.asarray(a[4], np.float32)))
    ref = []
    pipe2 = cm._RowPipeline(Blocking(), 64, lambda tag, data: ref.append(data))
    assert not pipe2.pipelined
    for tag, it in enumerate(items):
        pipe2.feed(it, tag)
    pipe2.flush(); pipe2.close()
    assert b"".join(d for _, d in got) == b"".join(ref) and b"".join(ref).count(b"\n") == n


def test_row_pipeline_surfaces_a_failing_sink():
    if not os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deepsignal_amd",
                                       "libdeepsignal_hip.so")):
        pytest.skip("native library not built (row formatter)")
    feats, items = _items_for_pipeline(200, [0, 100, 200])

    def sink(tag, data):
        raise IOError("disk full")
    pipe = cm._RowPipeline(_AsyncStandIn(), 64, sink)
    with pytest.raises(IOError):
        for tag, it in enumerate(items):
            pipe.feed(it, tag)
        pipe.flush()
    pipe.close()
