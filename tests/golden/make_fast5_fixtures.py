#!/opt/conda/bin/python3.9
"""Write the committed reads of extract_golden.json as REAL tombo-style single-read fast5 (HDF5) files with h5py.

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/make_fast5_fixtures.py

h5py exists only in this interpreter of the build container (not in the MI355X image), so the files are committed as
binary fixtures: tests/golden/fast5/*.fast5. They are data -- the arrays of extract_golden.json in the schema of
SURVEY.md Appendix C.4 -- in the HDF5 storage variants real files use:

  plain/   h5py defaults (superblock 0, symbol-table groups, contiguous datasets, fixed-length string attributes):
           what tombo's resquiggle writes with h5py;
  ont/     what MinKNOW-era files look like: Signal chunked + gzip + shuffle, Events chunked + gzip, read_id and the
           alignment strings as variable-length UTF-8 strings (global heap), a few extra groups / attributes around;
  latest/  `libver="latest"`-style object headers (version 2) and compact link-message groups with contiguous data.

Every file is read back with h5py AND with deepsignal_amd.minihdf5 here, and both must return the committed arrays."""
import json
import os
import sys

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from deepsignal_amd import minihdf5  # noqa: E402

G = json.load(open(os.path.join(HERE, "extract_golden.json")))
CG, SUB = "RawGenomeCorrected_000", "BaseCalled_template"


def write(path, r, style):
    sig = np.asarray(r["signal"], np.int16)
    rel = int(r["starts"][0])
    starts = np.asarray(r["starts"], np.int64) - rel
    lens = np.asarray(r["lengths"], np.int64)
    kw = {"libver": "latest"} if style == "latest" else {}
    ont = style == "ont"
    S = (lambda s: s) if ont else (lambda s: np.string_(s))          # Python str -> variable-length UTF-8 string attribute
    with h5py.File(path, "w", **kw) as f:
        if ont:
            f.attrs["file_version"] = 2.0
            ctx = f.create_group("UniqueGlobalKey/context_tags")
            ctx.attrs["experiment_type"] = "genomic_dna"
            trk = f.create_group("UniqueGlobalKey/tracking_id")
            trk.attrs["run_id"] = "0123456789abcdef"
            f.create_group("Analyses/Basecall_1D_000/BaseCalled_template")
        rd = f.create_group("Raw/Reads/Read_%d" % (17 + len(r["bases"])))
        if ont:
            rd.create_dataset("Signal", data=sig, chunks=(min(len(sig), 300),), compression="gzip", compression_opts=1, shuffle=True)
            rd.attrs["read_number"] = np.int32(17 + len(r["bases"]))
            rd.attrs["start_time"] = np.uint64(123456789)
            rd.attrs["duration"] = np.uint32(len(sig))
        else:
            rd.create_dataset("Signal", data=sig)
        rd.attrs["read_id"] = S(r["read_id"])
        ch = f.create_group("UniqueGlobalKey/channel_id")
        ch.attrs["digitisation"] = float(r["digitisation"])
        ch.attrs["range"] = float(r["range"])
        ch.attrs["offset"] = float(r["offset"])
        if ont:
            ch.attrs["sampling_rate"] = 4000.0
            ch.attrs["channel_number"] = "112"
        g = f.create_group("Analyses/%s/%s" % (CG, SUB))
        ev = np.zeros(len(lens), dtype=[("norm_mean", "<f8"), ("norm_stdev", "<f8"), ("start", "<u4"), ("length", "<u4"), ("base", "S1")])
        ev["start"], ev["length"], ev["base"] = starts, lens, [c.encode() for c in r["bases"]]
        if ont:
            d = g.create_dataset("Events", data=ev, chunks=(37,), compression="gzip")
        else:
            d = g.create_dataset("Events", data=ev)
        d.attrs["read_start_rel_to_raw"] = rel
        al = g.create_group("Alignment")
        al.attrs["mapped_strand"] = S(r["alignstrand"])
        al.attrs["mapped_chrom"] = S(r["chrom"])
        al.attrs["mapped_start"] = int(r["chrom_start"])
        if ont:
            al.attrs["mapped_end"] = int(r["chrom_start"]) + len(r["bases"])
            al.attrs["num_matches"] = np.int64(len(r["bases"]) - 3)


def read_with(mod, path):
    f = mod.File(path, "r")
    rd = list(f["Raw/Reads"].values())[0]
    dec = lambda v: v.decode() if isinstance(v, bytes) else str(v)
    ch = f["UniqueGlobalKey/channel_id"].attrs
    ev = f["Analyses/%s/%s/Events" % (CG, SUB)]
    al = f["Analyses/%s/%s/Alignment" % (CG, SUB)].attrs
    out = (np.asarray(rd["Signal"][()]).tolist(), dec(rd.attrs["read_id"]), float(ch["range"]), float(ch["digitisation"]), float(ch["offset"]),
           (np.asarray(ev["start"]).astype(np.int64) + int(ev.attrs["read_start_rel_to_raw"])).tolist(), np.asarray(ev["length"]).astype(np.int64).tolist(),
           "".join(b.decode() for b in ev["base"]), dec(al["mapped_strand"]), dec(al["mapped_chrom"]), int(al["mapped_start"]))
    f.close()
    return out


def main():
    n = 0
    for style in ("plain", "ont", "latest"):
        d = os.path.join(HERE, "fast5", style)
        os.makedirs(d, exist_ok=True)
        for name in G["read_order"]:
            r = G["reads"][name]
            p = os.path.join(d, name + ".fast5")
            write(p, r, style)
            want = (r["signal"], r["read_id"], r["range"], r["digitisation"], r["offset"], r["starts"], r["lengths"], r["bases"],
                    r["alignstrand"], r["chrom"], r["chrom_start"])
            assert read_with(h5py, p) == want, ("h5py", style, name)
            assert read_with(minihdf5, p) == want, ("minihdf5", style, name)
            n += 1
    print("wrote and cross-checked %d fast5 files; bytes:" % n,
          sum(os.path.getsize(os.path.join(dp, fn)) for dp, _, fns in os.walk(os.path.join(HERE, "fast5")) for fn in fns))


if __name__ == "__main__":
    main()
