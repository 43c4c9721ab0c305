"""CPU: scope row f2 — the from-scratch feature extractor against outputs of the REFERENCE extractor
(tests/golden/make_extract_golden.py ran deepsignal/extract_features.py on synthetic fast5 files; the
raw arrays of every read are committed, so no HDF5 library is needed here)."""
import json
import os
import random

import numpy as np
import pytest

from deepsignal_amd import extract_features as ef

GOLD = os.path.join(os.path.dirname(__file__), "golden", "extract_golden.json")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


@pytest.mark.parametrize("idx", [0, 1, 2, 3])
def test_feature_rows_match_reference(gold, idx):
    case = gold["cases"][idx]
    assert ef.get_motif_seqs(case["motifs"]) == case["motif_seqs"]
    random.seed(case["seed"])          # the >= 360-sample middle-base branch draws from `random`
    feats = []
    for name in gold["read_order"]:
        r = gold["reads"][name]
        chromlen = None if case["chrom2len"] is None else case["chrom2len"].get(r["chrom"])
        feats += ef.extract_read_features(
            np.asarray(r["signal"], np.int16), r["starts"], r["lengths"], r["bases"], r["range"] / r["digitisation"],
            r["offset"], r["read_id"], r["strand"], r["alignstrand"], r["chrom"], r["chrom_start"], chromlen,
            case["motif_seqs"], 0, case["kmer_len"], case["signal_len"], 1, case["normalize_method"])
    rows = [ef._features_to_str(f) for f in feats]
    assert rows == case["features_str"]               # byte-identical 12-column rows, in the same order
    for f, ref in zip(feats, case["features_head"]):
        assert [f[0], int(f[1]), f[2], int(f[3]), f[4], f[5], f[6]] == ref[:7]
        assert [float(x) for x in f[7]] == ref[7] and [float(x) for x in f[8]] == ref[8]      # un-rounded means / stds
        assert [int(x) for x in f[9]] == ref[9] and int(f[11]) == ref[11]


def test_rows_feed_the_call_mods_reader(gold, tmp_path):
    """extract -> TSV -> reader: the produced rows parse back to the same numbers (6-dp contract)."""
    from deepsignal_amd import call_modifications as cm
    case = gold["cases"][0]
    p = str(tmp_path / "features.tsv")
    with open(p, "w") as f:
        f.write("\n".join(case["features_str"]) + "\n")
    items = list(cm.iter_features_batches(p, 2))
    n = sum(len(it[0]) for it in items)
    assert n == len(case["features_str"])
    assert all(len(k) == 17 for it in items for k in it[1]) and all(len(s) == 360 for it in items for s in it[5])


def test_helpers():
    assert ef.get_motif_seqs("CG") == ["CG"] and sorted(ef.get_motif_seqs("CHG")) == ["CAG", "CCG", "CTG"]
    assert ef.get_refloc_of_methysite_in_motif("ACGTCGA", {"CG"}, 0) == [1, 4]
    x = np.array([1.0, 2.0, 4.0, 7.0, 100.0])
    assert abs(ef._mad(x) - np.median(np.abs(x - 4.0)) / 0.6744897501960817) < 1e-15
    with pytest.raises(ValueError):
        ef._normalize_signals(x, "nope")
    short = [np.arange(3.0), np.arange(4.0)]
    c = ef._get_central_signals(short, 10)
    assert list(c) == [0, 1, 2, 0, 1, 2, 3, 0, 0, 0]
    with pytest.raises(ValueError):
        ef.extract_read_features([0], [0], [1], "A", 1.0, 0.0, "r", "t", "+", "c", 0, None, ["CG"], 0, 16, 360, 1)


def _fake_reader(gold):
    def read(path, corrected_group, basecall_subgroup):
        r = gold["reads"][os.path.basename(path)[:-len(".fast5")]]
        return (np.asarray(r["signal"], np.int16), r["starts"], r["lengths"], r["bases"], r["range"] / r["digitisation"],
                r["offset"], (r["read_id"], r["strand"], r["alignstrand"], r["chrom"], r["chrom_start"]))
    return read


def test_extract_entry_point_and_cli(gold, tmp_path, monkeypatch):
    """`deepsignal extract` (extract_features.py:424-428 signature): fast5 directory -> feature TSV. HDF5 access is
    replaced by the committed raw arrays (no h5py in this image); the rows must be the reference extractor's rows, in one
    file or in a directory of <n>.tsv files, and a broken file is counted, not fatal."""
    from deepsignal_amd.deepsignal import main
    case = gold["cases"][0]
    d = tmp_path / "f5"
    d.mkdir()
    for name in gold["read_order"]:
        (d / (name + ".fast5")).write_bytes(b"")
    (d / "zzz_broken.fast5").write_bytes(b"")                  # unknown read -> the fake reader raises KeyError
    monkeypatch.setattr(ef, "_read_fast5", _fake_reader(gold))
    # files in the order the golden run processed them (its seeded `random.sample` draws depend on it), broken file last
    order = list(gold["read_order"])
    monkeypatch.setattr(ef, "get_fast5s", lambda fast5_dir, rec=True: [os.path.join(fast5_dir, n + ".fast5")
                                                                       for n in order + ["zzz_broken"]])
    want = list(case["features_str"])
    ref_fa = None
    if case["chrom2len"] is not None:                          # the golden run had a reference: contigs of these lengths
        ref_fa = str(tmp_path / "ref.fa")
        with open(ref_fa, "w") as f:
            for name, ln in case["chrom2len"].items():
                f.write(">%s some description\n" % name)
                for i in range(0, ln, 80):
                    f.write("A" * min(80, ln - i) + "\n")
        assert ef.read_reference_lengths(ref_fa) == case["chrom2len"]
    out = str(tmp_path / "features.tsv")
    random.seed(case["seed"])
    nrows, errors = ef.extract_features(str(d), True, ref_fa, True, 2, out, 1, "RawGenomeCorrected_000", "BaseCalled_template",
                                        case["normalize_method"], case["motifs"], 0, case["kmer_len"], case["signal_len"], 1,
                                        None, False, 200)
    got = open(out).read().splitlines()
    assert errors == 1 and nrows == len(got) == len(want)
    assert got == want                            # the reference extractor's rows, byte for byte, in order
    # directory output through the CLI: ceil(batches / w_batch_num) files
    outdir = str(tmp_path / "feat_dir")
    random.seed(case["seed"])
    assert main(["extract", "-i", str(d), "-o", outdir, "--w_is_dir", "yes", "--w_batch_num", "1", "--f5_batch_num", "2",
                 "--motifs", case["motifs"], "--normalize_method", case["normalize_method"],
                 "--kmer_len", str(case["kmer_len"]), "--cent_signals_len", str(case["signal_len"])]
                + (["--reference_path", ref_fa] if ref_fa else [])) == 0
    files = sorted(os.listdir(outdir), key=lambda f: int(f.split(".")[0]))
    nbatches = (len(order) + 1 + 1) // 2
    assert files == ["%d.tsv" % i for i in range(nbatches)]
    rows = [l for f in files for l in open(os.path.join(outdir, f)).read().splitlines()]
    assert rows == want


def test_extract_worker_pool_counts_failed_files(tmp_path):
    """--nproc > 1 spreads file batches over spawned workers; unreadable files (empty here, and no h5py in this image)
    are counted as failures, as the reference does (extract_features.py:225,281-283)."""
    d = tmp_path / "f5"
    d.mkdir()
    for i in range(5):
        (d / ("r%d.fast5" % i)).write_bytes(b"")
    out = str(tmp_path / "o.tsv")
    nrows, errors = ef.extract_features(str(d), True, None, True, 2, out, 2, "RawGenomeCorrected_000", "BaseCalled_template",
                                        "mad", "CG", 0, 17, 360, 1, None, False, 200)
    assert (nrows, errors) == (0, 5) and open(out).read() == ""
