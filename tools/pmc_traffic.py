"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> [out.json]

Units / corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB
and derive from the L2's memory-side request counters (Infinity-Cache hits are counted too);
on gfx950 FETCH_SIZE tallies the 128-B requests of wide (16 B/lane) coalesced reads at 64 B, so it is
doubled; WRITE_SIZE is exact for 16-B-per-lane stores.
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from build_identity import build_identity  # noqa: E402


def per_kernel(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv")
    acc = collections.OrderedDict()
    per_dispatch = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != counter or "ds::" not in r["Kernel_Name"]:
            continue
        did = int(r["Dispatch_Id"])
        per_dispatch[did] += float(r["Counter_Value"])
        names[did] = r["Kernel_Name"].split("(")[0].replace("void ds::", "").replace("ds::", "")
    for did, v in per_dispatch.items():
        a = acc.setdefault(names[did], [0, 0.0])
        a[0] += 1
        a[1] += v
    return {k: (n, s / n) for k, (n, s) in acc.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in fetch:
        n, fkib = fetch[k]
        wkib = write.get(k, (0, 0.0))[1]
        out[k] = {"dispatches": n,
                  "fetch_bytes_per_launch": fkib * 1024 * 2,     # gfx950 correction (x2)
                  "write_bytes_per_launch": wkib * 1024,
                  "hbm_bytes_per_launch": fkib * 2048 + wkib * 1024}
        print("%-40s n %5d  fetch %10.3f MB  write %10.3f MB" % (k, n, fkib * 2048 / 1e6, wkib * 1024 / 1e6))
    if len(sys.argv) > 3:
        json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py; "
                             "FETCH_SIZE KiB x2 (gfx950 wide-read correction), WRITE_SIZE KiB x1",
                   "kernels": out, **build_identity()}, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
