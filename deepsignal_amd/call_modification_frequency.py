"""Per-site modification frequency from call_mods result files — scope row f4 (the step after the
path; what users actually consume). Same algorithm, flags and output formats as the reference script
(/root/reference/scripts/call_modification_frequency.py:16-78, scripts/txt_formater.py:8-46):
group calls by (chromosome, pos), keep a call if |prob_0 - prob_1| >= prob_cf, accumulate prob sums,
met / unmet counts and coverage, write the 11-column table or bedMethyl.
"""
from __future__ import annotations

import argparse
import gzip
import os
import sys
from typing import Dict, Iterable, List, Tuple


class SiteStats:
    __slots__ = ("strand", "pos_in_strand", "kmer", "prob_0", "prob_1", "met", "unmet", "coverage")

    def __init__(self, strand: str, pos_in_strand: int, kmer: str):
        self.strand, self.pos_in_strand, self.kmer = strand, pos_in_strand, kmer
        self.prob_0 = 0.0
        self.prob_1 = 0.0
        self.met = self.unmet = self.coverage = 0


SiteKey = Tuple[str, int]


def calculate_mods_frequency(mods_files: Iterable[str], prob_cf: float = 0.0) -> Dict[SiteKey, SiteStats]:
    stats: Dict[SiteKey, SiteStats] = {}
    count = used = 0
    for path in mods_files:
        opener = gzip.open(path, "rt") if path.endswith(".gz") else open(path, "r")
        with opener as f:
            for line in f:
                w = line.strip().split("\t")
                prob_0, prob_1 = float(w[6]), float(w[7])
                count += 1
                if abs(prob_0 - prob_1) < prob_cf:
                    continue
                key = (w[0], int(w[1]))
                st = stats.get(key)
                if st is None:
                    st = stats[key] = SiteStats(w[2], int(w[3]), w[9])
                st.prob_0 += prob_0
                st.prob_1 += prob_1
                st.coverage += 1
                if int(w[8]) == 1:
                    st.met += 1
                else:
                    st.unmet += 1
                used += 1
    print("{:.2f}% ({} of {}) calls used..".format(used / float(count) * 100 if count else 0.0, used, count))
    return stats


def write_sitekey2stats(stats: Dict[SiteKey, SiteStats], result_file: str, is_sort: bool, is_bed: bool) -> None:
    keys: List[SiteKey] = sorted(stats) if is_sort else list(stats)
    with open(result_file, "w") as wf:
        for chrom, pos in keys:
            st = stats[(chrom, pos)]
            assert st.coverage == st.met + st.unmet
            if st.coverage <= 0:
                print("{} {} has no coverage..".format(chrom, pos))
                continue
            rmet = float(st.met) / st.coverage
            if is_bed:
                wf.write("\t".join([chrom, str(pos), str(pos + 1), ".", str(st.coverage), st.strand, str(pos),
                                    str(pos + 1), "0,0,0", str(st.coverage), str(int(round(rmet * 100, 0)))]) + "\n")
            else:
                wf.write("%s\t%d\t%s\t%d\t%.3f\t%.3f\t%d\t%d\t%d\t%.4f\t%s\n" % (
                    chrom, pos, st.strand, st.pos_in_strand, st.prob_0, st.prob_1, st.met, st.unmet, st.coverage,
                    rmet, st.kmer))


def collect_input_files(input_paths: List[str], file_uid=None) -> List[str]:
    files = []
    for ipath in input_paths:
        p = os.path.abspath(ipath)
        if os.path.isdir(p):
            for name in os.listdir(p):
                if file_uid is None or name.find(file_uid) != -1:
                    files.append("/".join([p, name]))
        elif os.path.isfile(p):
            files.append(p)
        else:
            raise ValueError("%s is neither a file nor a directory" % ipath)
    return files


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description="calculate frequency of interested sites at genome level")
    ap.add_argument("--input_path", "-i", action="append", type=str, required=True)
    ap.add_argument("--result_file", "-o", type=str, required=True)
    ap.add_argument("--bed", action="store_true", default=False)
    ap.add_argument("--sort", action="store_true", default=False)
    ap.add_argument("--prob_cf", type=float, default=0.0)
    ap.add_argument("--file_uid", type=str, default=None)
    a = ap.parse_args(argv)
    files = collect_input_files(a.input_path, a.file_uid)
    print("get {} input file(s)..".format(len(files)))
    write_sitekey2stats(calculate_mods_frequency(files, a.prob_cf), a.result_file, a.sort, a.bed)
    return 0


if __name__ == "__main__":
    sys.exit(main())
