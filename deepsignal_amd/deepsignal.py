"""`deepsignal call_mods` command line — the reference's flag surface for this sub-command
(reference deepsignal/deepsignal.py:236-326, defaults included), driving the MI355X engine — plus `extract`, the
host-side step that produces call_mods' feature-TSV input (deepsignal.py:155-234).

Multi-GPU: one process per GPU, e.g.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        -m deepsignal_amd.deepsignal call_mods -i features.tsv -m model.ckpt -o calls.tsv
reads are dealt to the ranks, rank 0 gathers the results over RCCL and writes the file."""
from __future__ import absolute_import

import argparse
import sys

from .utils.process_utils import display_args, str2bool


def main_call_mods(args):
    from .call_modifications import call_mods
    display_args(args)
    # the reference's tuple, in the reference's order (deepsignal.py:83-84); methy_label is fixed to 1 there (:78-79)
    f5_args = (str2bool(args.recursively), args.corrected_group, args.basecall_subgroup, args.reference_path,
               str2bool(args.is_dna), args.normalize_method, args.motifs, args.mod_loc, 1, args.f5_batch_num,
               args.positions)
    call_mods(args.input_path, args.model_path, args.result_file, args.kmer_len, args.cent_signals_len,
              args.batch_size, args.learning_rate, args.class_num, args.nproc, str2bool(args.is_gpu),
              str2bool(args.is_rnn), str2bool(args.is_base), str2bool(args.is_cnn), f5_args,
              precision=args.precision, engine_batch=args.engine_batch)


def main_extraction(args):
    from .extract_features import extract_features
    display_args(args)
    extract_features(args.fast5_dir, str2bool(args.recursively), args.reference_path, str2bool(args.is_dna),
                     args.f5_batch_num, args.write_path, args.nproc, args.corrected_group, args.basecall_subgroup,
                     args.normalize_method, args.motifs, args.mod_loc, args.kmer_len, args.cent_signals_len,
                     args.methy_label, args.positions, str2bool(args.w_is_dir), args.w_batch_num)


def build_parser():
    parser = argparse.ArgumentParser(prog="deepsignal", description="call_mods on MI355X (gfx950)")
    sub = parser.add_subparsers(title="modules", dest="module")
    # `extract`: the step before the path -- fast5 -> feature TSV (reference deepsignal/deepsignal.py:155-234, same flags)
    e = sub.add_parser("extract", description="extract features from fast5 files (host side; HDF5 through h5py, or deepsignal_amd.minihdf5 where h5py is absent)")
    g = e.add_argument_group("INPUT")
    g.add_argument("--fast5_dir", "-i", required=True)
    g.add_argument("--recursively", "-r", default="yes")
    g.add_argument("--corrected_group", default="RawGenomeCorrected_000")
    g.add_argument("--basecall_subgroup", default="BaseCalled_template")
    g.add_argument("--is_dna", default="yes")
    g.add_argument("--reference_path", default=None)
    g = e.add_argument_group("EXTRACTION")
    g.add_argument("--normalize_method", default="mad", choices=["mad", "zscore"])
    g.add_argument("--methy_label", type=int, default=1, choices=[1, 0])
    g.add_argument("--kmer_len", type=int, default=17)
    g.add_argument("--cent_signals_len", type=int, default=360)
    g.add_argument("--motifs", default="CG")
    g.add_argument("--mod_loc", type=int, default=0)
    g.add_argument("--positions", default=None)
    g = e.add_argument_group("OUTPUT")
    g.add_argument("--write_path", "-o", required=True)
    g.add_argument("--w_is_dir", default="no")
    g.add_argument("--w_batch_num", type=int, default=200)
    e.add_argument("--nproc", "-p", type=int, default=1)
    e.add_argument("--f5_batch_num", type=int, default=50)
    e.set_defaults(func=main_extraction)
    p = sub.add_parser("call_mods", description="call modifications")
    g = p.add_argument_group("INPUT")
    g.add_argument("--input_path", "-i", required=True,
                   help="a file of extracted features (fast5 dirs need `extract` first)")
    g.add_argument("--f5_batch_num", type=int, default=50)
    g = p.add_argument_group("CALL")
    g.add_argument("--model_path", "-m", required=True,
                   help="TensorFlow checkpoint prefix of a reference-trained model (<prefix>.index + .data-*), "
                        "or a DSAMDW01 weight file")
    g.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3", "bf16", "bf16_all"],
                   help="fp32 (reference numerics, native fp32 matrix instructions); bf16x3 (fp32-class results: fp32 operands carried "
                        "as three bf16 terms on the bf16 matrix pipe, held to the fp32 parity bars); bf16 / bf16_all (bf16 conv+FC "
                        "operands with fp32 accumulate: fast, probabilities good to ~1e-2 only)")
    g.add_argument("--engine_batch", type=int, default=0,
                   help="sites per GPU forward the engine is created for (default 0: the larger of --batch_size and 4096; results do "
                        "not depend on it, device and pinned memory grow with it -- lower it on a small or shared GPU; the "
                        "environment variable DS_ENGINE_BATCH does the same)")
    g.add_argument("--is_cnn", default="yes")
    g.add_argument("--is_rnn", default="yes")
    g.add_argument("--is_base", default="yes")
    g.add_argument("--kmer_len", "-x", type=int, default=17)
    g.add_argument("--cent_signals_len", "-y", type=int, default=360)
    g.add_argument("--batch_size", "-b", type=int, default=512,
                   help="the reference's rows per sess.run; here the granularity rows are handed to the engine in, NOT the sites per "
                        "GPU forward (see --engine_batch): a site's result does not depend on its batch mates")
    g.add_argument("--learning_rate", "-l", type=float, default=0.001)
    g.add_argument("--class_num", "-c", type=int, default=2)
    g = p.add_argument_group("OUTPUT")
    g.add_argument("--result_file", "-o", required=True)
    g = p.add_argument_group("EXTRACTION")
    g.add_argument("--recursively", "-r", default="yes")
    g.add_argument("--corrected_group", default="RawGenomeCorrected_000")
    g.add_argument("--basecall_subgroup", default="BaseCalled_template")
    g.add_argument("--is_dna", default="yes")
    g.add_argument("--normalize_method", default="mad", choices=["mad", "zscore"])
    g.add_argument("--motifs", default="CG")
    g.add_argument("--mod_loc", type=int, default=0)
    g.add_argument("--positions", default=None)
    g.add_argument("--reference_path", default=None)
    p.add_argument("--nproc", "-p", type=int, default=1)
    p.add_argument("--is_gpu", default="no", choices=["yes", "no"])
    p.set_defaults(func=main_call_mods)
    return parser


def main(argv=None):
    parser = build_parser()
    args = parser.parse_args(argv)
    if not getattr(args, "func", None):
        parser.print_help()
        return 1
    args.func(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
