"""fast5 -> per-site features (scope row f2; stays host Python as the north star asks).

From-scratch statement of the reference extractor
(/root/reference/deepsignal/extract_features.py:35-72,121-286,289-303 and
utils/process_utils.py:95-143): raw pA rescale -> MAD / z-score normalisation rounded to 6 dp ->
per-base event slices -> motif scan -> per-site k-mer, means, stds, lengths and the 360 central
signal samples -> the 12-column feature row.

The numeric core works on plain arrays (`extract_read_features`), so it is testable without HDF5;
tombo-resquiggled single-read fast5 files are opened with `h5py` where it exists and with deepsignal_amd.minihdf5
(plain Python, read-only) where it does not -- the MI355X image.
"""
from __future__ import annotations

import os
import sys
import random
from typing import Dict, Iterable, List, Optional, Sequence, Set, Tuple

import numpy as np

key_sep = "||"
MAD_NORMAL_CONSISTENCY = 0.6744897501960817    # Phi^-1(3/4): statsmodels.robust.mad's default scale

iupac_alphabets = {"A": ["A"], "T": ["T"], "C": ["C"], "G": ["G"], "R": ["A", "G"], "M": ["A", "C"], "S": ["C", "G"],
                   "Y": ["C", "T"], "K": ["G", "T"], "W": ["A", "T"], "B": ["C", "G", "T"], "D": ["A", "G", "T"],
                   "H": ["A", "C", "T"], "V": ["A", "C", "G"], "N": ["A", "C", "G", "T"]}
iupac_alphabets_rna = {k: [("U" if b == "T" else b) for b in v] for k, v in iupac_alphabets.items() if k != "T"}
iupac_alphabets_rna["U"] = ["U"]


_UNSUPPORTED_SEEN = set()      # minihdf5.Unsupported messages already reported (per process)

def get_motif_seqs(motifs: str, is_dna: bool = True) -> List[str]:
    """'CG' / 'CHG,CHH' -> every concrete sequence of the IUPAC motifs (process_utils.py:113-143)."""
    table = iupac_alphabets if is_dna else iupac_alphabets_rna
    out: List[str] = []
    for motif in motifs.strip().split(","):
        seqs = [""]
        for ch in motif.strip().upper():
            seqs = [s + b for s in seqs for b in table[ch]]
        out += seqs
    return out


def get_refloc_of_methysite_in_motif(seqstr: str, motifset: Iterable[str], methyloc_in_motif: int = 0) -> List[int]:
    """0-based positions of the targeted base of every motif hit (process_utils.py:95-110)."""
    motifset = set(motifset)
    motiflen = len(next(iter(motifset)))
    return [i + methyloc_in_motif for i in range(0, len(seqstr) - motiflen + 1) if seqstr[i:i + motiflen] in motifset]


def _rescale_signals(rawsignals, scaling: float, offset: float) -> np.ndarray:
    return np.array(scaling * (np.asarray(rawsignals) + offset), dtype=np.float64)     # extract_features.py:211-212


def _mad(x: np.ndarray) -> float:
    return float(np.median(np.abs(x - np.median(x))) / MAD_NORMAL_CONSISTENCY)


def _normalize_signals(signals: np.ndarray, normalize_method: str = "mad") -> np.ndarray:
    if normalize_method == "zscore":
        sshift, sscale = np.mean(signals), float(np.std(signals))
    elif normalize_method == "mad":
        sshift, sscale = np.median(signals), _mad(signals)
    else:
        raise ValueError("normalize_method must be 'mad' or 'zscore'")
    return np.around((signals - sshift) / sscale, decimals=6)          # extract_features.py:143-151


def _get_central_signals(signals_list: Sequence[np.ndarray], rawsignal_num: int = 360):
    """The rawsignal_num samples centred on the middle base (extract_features.py:154-190)."""
    total = sum(len(x) for x in signals_list)
    if total < rawsignal_num:
        real = np.concatenate(signals_list)
        return np.append(real, np.array([0] * (rawsignal_num - len(real))))
    mid_loc = int((len(signals_list) - 1) / 2)
    mid = signals_list[mid_loc]
    if len(mid) >= rawsignal_num:       # ordered random subsample of the (very long) middle base
        return [mid[x] for x in sorted(random.sample(range(len(mid)), rawsignal_num))]
    left_len = (rawsignal_num - len(mid)) // 2
    right_len = rawsignal_num - left_len
    left = np.concatenate(signals_list[:mid_loc])
    right = np.concatenate(signals_list[mid_loc:])
    if left_len > len(left):
        right_len += left_len - len(left)
        left_len = len(left)
    elif right_len > len(right):
        left_len += right_len - len(right)
        right_len = len(right)
    assert right_len + left_len == rawsignal_num
    if left_len == 0:
        return right[:right_len]
    return np.append(left[-left_len:], right[:right_len])


def extract_read_features(raw_signal, starts, lengths, bases: str, scaling: float, offset: float, readname: str,
                          strand: str, alignstrand: str, chrom: str, chrom_start: int, chromlen: Optional[int],
                          motif_seqs: Iterable[str], methyloc: int, kmer_len: int, raw_signals_len: int,
                          methy_label: int, normalize_method: str = "mad", positions: Optional[Set[str]] = None):
    """All feature tuples of one read (the body of the reference's per-file loop, :225-280)."""
    if kmer_len % 2 == 0:
        raise ValueError("kmer_len must be odd")
    num_bases = (kmer_len - 1) // 2
    norm = _normalize_signals(_rescale_signals(raw_signal, scaling, offset), normalize_method)
    signal_list = [norm[int(s):int(s) + int(l)] for s, l in zip(starts, lengths)]
    genomeseq = bases
    out = []
    for loc in get_refloc_of_methysite_in_motif(genomeseq, set(motif_seqs), methyloc):
        if not (num_bases <= loc < len(genomeseq) - num_bases):
            continue
        if alignstrand == "-":
            pos = chrom_start + len(genomeseq) - 1 - loc
            pos_in_strand = chromlen - 1 - pos if chromlen is not None else -1
        else:
            pos = chrom_start + loc
            pos_in_strand = pos if chromlen is not None else -1
        if positions is not None and key_sep.join([chrom, str(pos), alignstrand]) not in positions:
            continue
        k_mer = genomeseq[loc - num_bases:loc + num_bases + 1]
        k_signals = signal_list[loc - num_bases:loc + num_bases + 1]
        signal_lens = [len(x) for x in k_signals]
        signal_means = [np.mean(x) for x in k_signals]
        signal_stds = [np.std(x) for x in k_signals]
        cent = _get_central_signals(k_signals, raw_signals_len)
        out.append((chrom, pos, alignstrand, pos_in_strand, readname, strand, k_mer, signal_means, signal_stds,
                    signal_lens, cent, methy_label))
    return out


def _features_to_str(features) -> str:
    """One 12-column feature row (extract_features.py:289-303)."""
    chrom, pos, alignstrand, pos_in_strand, readname, strand, k_mer, means, stds, lens, cent, label = features
    return "\t".join([chrom, str(pos), alignstrand, str(pos_in_strand), readname, strand, k_mer,
                      ",".join(str(x) for x in np.around(means, decimals=6)),
                      ",".join(str(x) for x in np.around(stds, decimals=6)),
                      ",".join(str(x) for x in lens), ",".join(str(x) for x in cent), str(label)])


# ------------------------------------------------------------------ fast5 (HDF5) access: h5py, or deepsignal_amd.minihdf5
def _hdf5_module():
    """h5py where it is installed (as the reference uses, extract_features.py:8), else the package's own read-only HDF5
    reader (deepsignal_amd/minihdf5.py: the MI355X image has no h5py). Both open the real file; tests/test_minihdf5.py holds
    them to the same arrays on the committed fast5 fixtures."""
    try:
        import h5py
        return h5py
    except ImportError:
        from . import minihdf5
        return minihdf5


def _read_fast5(path: str, corrected_group: str, basecall_subgroup: str, hdf5=None):
    """Schema of SURVEY.md Appendix C.4 (extract_features.py:35-72,75-140,193-208)."""
    h5py = hdf5 if hdf5 is not None else _hdf5_module()
    with h5py.File(path, "r") as f:
        read = list(f["Raw/Reads"].values())[0]
        raw = read["Signal"][()]
        rid = read.attrs["read_id"]
        rid = rid.decode() if isinstance(rid, bytes) else str(rid)
        ch = f["UniqueGlobalKey/channel_id"].attrs
        scaling, offset = ch["range"] / ch["digitisation"], ch["offset"]
        base = "/".join(["Analyses", corrected_group, basecall_subgroup])
        ev = f[base + "/Events"]
        rel = ev.attrs["read_start_rel_to_raw"]
        starts = ev["start"].astype(np.int64) + int(rel)
        lengths = ev["length"].astype(np.int64)
        bases = "".join(b.decode("UTF-8") for b in ev["base"])
        if base + "/Alignment" in f:
            al = f[base + "/Alignment"].attrs
            dec = lambda v: v.decode() if isinstance(v, bytes) else str(v)
            info = (rid, "t" if basecall_subgroup.endswith("template") else "c", dec(al["mapped_strand"]),
                    dec(al["mapped_chrom"]), int(al["mapped_start"]))
        else:
            info = ("", "", "", "", "")
    return raw, starts, lengths, bases, scaling, offset, info


def _extract_features(fast5s, corrected_group, basecall_subgroup, normalize_method, motif_seqs, methyloc, chrom2len,
                      kmer_len, raw_signals_len, methy_label, positions):
    """Reference signature (extract_features.py:215-286): -> (features_list, number of failed files)."""
    if kmer_len % 2 == 0:
        raise ValueError("kmer_len must be odd")
    features_list, error = [], 0
    for fp in fast5s:
        try:
            raw, starts, lengths, bases, scaling, offset, info = _read_fast5(fp, corrected_group, basecall_subgroup)
            readname, strand, alignstrand, chrom, chrom_start = info
            chromlen = None
            if chrom2len is not None:
                chromlen = chrom2len.get(chrom)
                if chromlen is None:
                    print("warning - chrom_name in fast5 not in provided reference genome!")
            features_list += extract_read_features(raw, starts, lengths, bases, scaling, offset, readname, strand,
                                                   alignstrand, chrom, chrom_start, chromlen, motif_seqs, methyloc,
                                                   kmer_len, raw_signals_len, methy_label, normalize_method, positions)
        except Exception as exc:
            # the reference counts and continues (extract_features.py:263-265). A file the built-in HDF5 reader cannot
            # open for a NAMED reason (VBZ-compressed signal, dense groups ...) is also counted, but the reason is
            # printed once per kind: a directory of such files would otherwise end as "N failed" and nothing else
            error += 1
            if type(exc).__name__ == "Unsupported":
                msg = str(exc)
                if msg not in _UNSUPPORTED_SEEN:
                    _UNSUPPORTED_SEEN.add(msg)
                    print("warning - %s: fast5 feature not supported by the built-in HDF5 reader: %s (install h5py to read "
                          "such files; further files failing for this reason are only counted)" % (fp, msg), file=sys.stderr)
    return features_list, error


def get_fast5s(fast5_dir: str, is_recursive: bool = True) -> List[str]:
    fast5_dir = os.path.abspath(fast5_dir)
    if is_recursive:
        return [os.path.join(r, f) for r, _, fs in os.walk(fast5_dir) for f in fs if f.endswith(".fast5")]
    return ["/".join([fast5_dir, f]) for f in os.listdir(fast5_dir) if f.endswith(".fast5")]


def read_reference_lengths(reference_path: Optional[str]):
    """Contig name -> length of a FASTA file, or None (reference utils/ref_reader.py:7-13 role)."""
    if reference_path is None:
        return None
    chrom2len, name = {}, None
    with open(reference_path) as rf:
        for line in rf:
            if line.startswith(">"):
                name = line.strip()[1:].split(" ")[0]
                chrom2len[name] = 0
            elif name is not None:
                chrom2len[name] += len(line.strip())
    return chrom2len


def read_position_file(position_file: Optional[str]):
    """chromosome \t position \t strand rows -> set of keys, or None (extract_features.py:388-394 role)."""
    if position_file is None:
        return None
    with open(position_file) as pf:
        return set(key_sep.join(line.strip().split("\t")[:3]) for line in pf)


def _extract_batch(task):
    (fast5s, corrected_group, basecall_subgroup, normalize_method, motif_seqs, methyloc, chrom2len, kmer_len,
     raw_signals_len, methy_label, positions) = task
    feats, err = _extract_features(fast5s, corrected_group, basecall_subgroup, normalize_method, motif_seqs, methyloc,
                                   chrom2len, kmer_len, raw_signals_len, methy_label, positions)
    return [_features_to_str(f) for f in feats], err


class _FeatureWriter:
    """One file, or a directory of <n>.tsv files holding w_batch_num batches each (extract_features.py:336-385 role)."""

    def __init__(self, write_fp: str, is_dir: bool, w_batch_num: int):
        self.is_dir, self.w_batch_num, self.dir = is_dir, max(1, w_batch_num), write_fp
        self.file_count = self.batch_count = 0
        if is_dir:
            if os.path.isfile(write_fp):
                raise FileExistsError("{} already exists as a file, please use another write_dir".format(write_fp))
            os.makedirs(write_fp, exist_ok=True)
            self.wf = open("/".join([write_fp, "0.tsv"]), "w")
        else:
            self.wf = open(write_fp, "w")

    def write_batch(self, rows):
        if self.is_dir and self.batch_count >= self.w_batch_num:
            self.wf.close()
            self.file_count += 1
            self.batch_count = 0
            self.wf = open("/".join([self.dir, str(self.file_count) + ".tsv"]), "w")
        for row in rows:
            self.wf.write(row + "\n")
        self.wf.flush()
        self.batch_count += 1

    def close(self):
        self.wf.close()


def extract_features(fast5_dir, is_recursive, reference_path, is_dna, batch_size, write_fp, nproc,
                     corrected_group, basecall_subgroup, normalize_method, motifs, methyloc, kmer_len, raw_signals_len,
                     methy_label, position_file, w_is_dir, w_batch_num):
    """`deepsignal extract`: fast5 directory -> feature TSV (same arguments as the reference's
    extract_features.py:424-428). Files are taken in batches of `batch_size`; `nproc` > 1 spreads the batches over a
    process pool (rows of a batch stay together; batches are written in completion order, as in the reference)."""
    import time
    start = time.time()
    fast5s = get_fast5s(fast5_dir, is_recursive)
    print("{} fast5 files in total..".format(len(fast5s)))
    motif_seqs = get_motif_seqs(motifs, is_dna)
    chrom2len = read_reference_lengths(reference_path)
    positions = read_position_file(position_file)
    tasks = [(fast5s[i:i + batch_size], corrected_group, basecall_subgroup, normalize_method, motif_seqs, methyloc,
              chrom2len, kmer_len, raw_signals_len, methy_label, positions) for i in range(0, len(fast5s), batch_size)]
    writer = _FeatureWriter(write_fp, w_is_dir, w_batch_num)
    errors = nrows = 0
    try:
        if nproc > 1 and len(tasks) > 1:
            import multiprocessing as mp
            with mp.get_context("spawn").Pool(min(nproc, len(tasks))) as pool:
                for rows, err in pool.imap_unordered(_extract_batch, tasks):
                    writer.write_batch(rows)
                    errors += err
                    nrows += len(rows)
        else:
            for task in tasks:
                rows, err = _extract_batch(task)
                writer.write_batch(rows)
                errors += err
                nrows += len(rows)
    finally:
        writer.close()
    print("%d of %d fast5 files failed.." % (errors, len(fast5s)))
    print("[extract] finished, cost {:.1f}s ({} feature rows)".format(time.time() - start, nrows))
    return nrows, errors

