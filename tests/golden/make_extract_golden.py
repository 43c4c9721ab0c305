#!/opt/conda/bin/python3.9
"""Golden vectors for scope row f2 (fast5 -> features) by RUNNING the reference extractor.

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/make_extract_golden.py

Builds synthetic tombo-style single-read fast5 files with h5py (schema: SURVEY.md Appendix C.4), runs
/root/reference/deepsignal/extract_features.py::_extract_features / _features_to_str on them
(tensorflow is not needed by that module; the removed numpy aliases np.int / np.float are shimmed) and
commits the RAW ARRAYS of every read plus the reference's outputs (data only). h5py exists only in
this interpreter, so the committed arrays are what the tests feed to the from-scratch extractor."""
import json
import os
import random
import sys
import tempfile

import h5py
import numpy as np

np.int = int
np.float = float
sys.path.insert(0, "/root/reference")
from deepsignal import extract_features as ref   # noqa: E402
from deepsignal.utils.process_utils import get_motif_seqs   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(11)


def make_read(path, nbases, strand, chrom, start, mean_len=9, special=None):
    bases = "".join("ACGT"[i] for i in rng.integers(0, 4, nbases))
    # sprinkle CG motifs
    b = list(bases)
    for i in range(12, nbases - 12, 7):
        b[i], b[i + 1] = "C", "G"
    bases = "".join(b)
    lens = (1 + rng.poisson(mean_len - 1, nbases)).astype(np.int64)
    if special == "long_mid":
        lens[40] = 420
    if special == "short":
        lens[:] = 3
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    rel = int(rng.integers(5, 50))
    total = int(rel + lens.sum() + 20)
    signal = (rng.normal(500, 60, total)).astype(np.int16)
    with h5py.File(path, "w") as f:
        rd = f.create_group("Raw/Reads/Read_%d" % rng.integers(1, 1000))
        rd.create_dataset("Signal", data=signal)
        rd.attrs["read_id"] = np.string_("read-%s" % os.path.basename(path)[:-6])
        ch = f.create_group("UniqueGlobalKey/channel_id")
        ch.attrs["digitisation"] = 8192.0
        ch.attrs["range"] = 1467.61
        ch.attrs["offset"] = float(rng.integers(-10, 30))
        g = f.create_group("Analyses/RawGenomeCorrected_000/BaseCalled_template")
        ev = np.zeros(nbases, dtype=[("norm_mean", "<f8"), ("norm_stdev", "<f8"), ("start", "<u4"), ("length", "<u4"), ("base", "S1")])
        ev["start"], ev["length"], ev["base"] = starts, lens, [c.encode() for c in bases]
        d = g.create_dataset("Events", data=ev)
        d.attrs["read_start_rel_to_raw"] = rel
        al = g.create_group("Alignment")
        al.attrs["mapped_strand"] = np.string_(strand)
        al.attrs["mapped_chrom"] = np.string_(chrom)
        al.attrs["mapped_start"] = start
    return {"signal": signal.tolist(), "starts": (starts + rel).tolist(), "lengths": lens.tolist(), "bases": bases,
            "digitisation": 8192.0, "range": 1467.61, "offset": float(h5py.File(path, "r")["UniqueGlobalKey/channel_id"].attrs["offset"]),
            "read_id": "read-%s" % os.path.basename(path)[:-6], "strand": "t", "alignstrand": strand, "chrom": chrom,
            "chrom_start": int(start)}


def main():
    cases = []
    with tempfile.TemporaryDirectory() as d:
        specs = [("a", 120, "+", "chr1", 1000, 9, None), ("b", 90, "-", "chr2", 5000, 9, None),
                 ("c", 100, "+", "chr1", 20, 9, "long_mid"), ("d", 80, "-", "chr1", 300, 3, "short"),
                 ("e", 70, "+", "chr3", 0, 30, None)]
        reads = {}
        for name, nb, st, ch, start, ml, sp in specs:
            p = os.path.join(d, name + ".fast5")
            reads[name] = make_read(p, nb, st, ch, start, ml, sp)
        files = [os.path.join(d, s[0] + ".fast5") for s in specs]
        chrom2len = {"chr1": 100000, "chr2": 200000, "chr3": 5000}
        for cname, norm, motifs, kmer, siglen, c2l, positions in [
                ("mad_cg", "mad", "CG", 17, 360, chrom2len, None),
                ("zscore_cg_noref", "zscore", "CG", 17, 360, None, None),
                ("mad_k9_sig100", "mad", "CG", 9, 100, chrom2len, None),
                ("mad_iupac_chg", "mad", "CHG", 17, 360, chrom2len, None)]:
            random.seed(1234)
            motif_seqs = get_motif_seqs(motifs, True)
            feats, err = ref._extract_features(files, "RawGenomeCorrected_000", "BaseCalled_template", norm, motif_seqs, 0,
                                               c2l, kmer, siglen, 1, positions)
            strs = [ref._features_to_str(f) for f in feats]
            cases.append({"name": cname, "normalize_method": norm, "motifs": motifs, "motif_seqs": motif_seqs, "kmer_len": kmer,
                          "signal_len": siglen, "chrom2len": c2l, "error": err, "seed": 1234,
                          "features_str": strs,
                          "features_head": [[f[0], int(f[1]), f[2], int(f[3]), f[4], f[5], f[6], [float(x) for x in f[7]],
                                             [float(x) for x in f[8]], [int(x) for x in f[9]], [float(x) for x in f[10]][:8], int(f[11])]
                                            for f in feats[:40]]})
        with open(os.path.join(HERE, "extract_golden.json"), "w") as f:
            json.dump({"generator": "tests/golden/make_extract_golden.py (reference extractor on synthetic fast5s)",
                       "read_order": [s[0] for s in specs], "reads": reads, "cases": cases}, f)
    print("wrote extract_golden.json:", [(c["name"], len(c["features_str"]), c["error"]) for c in cases])


if __name__ == "__main__":
    main()
