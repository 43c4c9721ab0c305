"""deepsignal_amd.minihdf5 on REAL HDF5 files: tests/golden/fast5/{plain,ont,latest}/*.fast5 were written by h5py
(tests/golden/make_fast5_fixtures.py, which also checked h5py's own read-back against the same arrays) from the reads of
extract_golden.json -- the arrays the REFERENCE extractor was run on. So: files -> this reader -> from-scratch extractor
must reproduce the reference's feature rows byte for byte, in every storage variant (contiguous / chunked + gzip +
shuffle, fixed / variable-length strings, symbol-table / link-message groups, object headers v1 / v2)."""
import json
import os
import random

import numpy as np
import pytest

from deepsignal_amd import extract_features as ef, minihdf5

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F5 = os.path.join(ROOT, "tests", "golden", "fast5")
G = json.load(open(os.path.join(ROOT, "tests", "golden", "extract_golden.json")))
CG, SUB = "RawGenomeCorrected_000", "BaseCalled_template"
STYLES = ("plain", "ont", "latest")


@pytest.mark.parametrize("style", STYLES)
def test_read_fast5_returns_the_committed_arrays(style):
    for name in G["read_order"]:
        r = G["reads"][name]
        raw, starts, lengths, bases, scaling, offset, info = ef._read_fast5(os.path.join(F5, style, name + ".fast5"), CG, SUB, hdf5=minihdf5)
        assert raw.dtype == np.int16 and raw.tolist() == r["signal"]
        assert starts.tolist() == r["starts"] and lengths.tolist() == r["lengths"] and bases == r["bases"]
        assert scaling == r["range"] / r["digitisation"] and offset == r["offset"]
        assert info == (r["read_id"], "t", r["alignstrand"], r["chrom"], r["chrom_start"])


@pytest.mark.parametrize("style", STYLES)
@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_extractor_on_real_files_prints_the_references_rows(style, case, monkeypatch):
    c = G["cases"][case]
    monkeypatch.setattr(ef, "_hdf5_module", lambda: minihdf5)
    files = [os.path.join(F5, style, n + ".fast5") for n in G["read_order"]]
    random.seed(c["seed"])
    feats, err = ef._extract_features(files, CG, SUB, c["normalize_method"], c["motif_seqs"], 0, c["chrom2len"], c["kmer_len"],
                                      c["signal_len"], 1, None)
    assert err == c["error"]
    assert [ef._features_to_str(f) for f in feats] == c["features_str"]


def test_api_subset_and_loud_failures(tmp_path):
    f = minihdf5.File(os.path.join(F5, "ont", "a.fast5"))
    assert sorted(f.keys()) == ["Analyses", "Raw", "UniqueGlobalKey"]
    assert "Analyses/%s/%s/Alignment" % (CG, SUB) in f and "Analyses/nothing" not in f
    rd = list(f["Raw/Reads"].values())[0]
    sig = rd["Signal"]
    assert sig.shape == (len(G["reads"]["a"]["signal"]),) and sig.dtype == np.int16 and sig[:5].tolist() == G["reads"]["a"]["signal"][:5]
    assert rd.attrs["read_id"] == "read-a" and "read_number" in rd.attrs and int(rd.attrs["read_number"]) == 17 + len(G["reads"]["a"]["bases"])
    assert f.attrs["file_version"] == 2.0
    ev = f["Analyses/%s/%s/Events" % (CG, SUB)]
    assert ev.dtype.names == ("norm_mean", "norm_stdev", "start", "length", "base") and len(ev) == len(G["reads"]["a"]["bases"])
    with pytest.raises(KeyError):
        f["Raw/Reads/none"]
    bad = tmp_path / "x.fast5"
    bad.write_bytes(b"not hdf5 at all" * 100)
    with pytest.raises(ValueError):
        minihdf5.File(str(bad))
    # a recent ONT file compresses the signal with the VBZ plugin filter: named, not mis-read
    ds = rd["Signal"]
    with pytest.raises(minihdf5.Unsupported, match="VBZ"):
        ds._unfilter(b"\x00" * 8, 0, [(32020, [])])
