"""call modifications from extracted feature files with the MI355X engine.

Host-side mirror of the reference's batch harness (deepsignal/call_modifications.py): the same
function names, argument meaning, queue-item layout, "kill" sentinel and output row text — with
`(tf_sess, model)` replaced by a `deepsignal_amd.engine.Engine` (anything with a
`run(kmer, means, stds, sanums, signals) -> (activation_logits, prediction)` method).

output format: chromosome, pos, strand, pos_in_strand, read_name, read_strand,
prob_0, prob_1, called_label, seq          (reference call_modifications.py:4-5, README.md:170-180)
"""
from __future__ import absolute_import

import os
import sys
import time
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np

from .utils.process_utils import base2code_dna, code2base_dna, str2bool

# a queue item, exactly as the reference builds it (call_modifications.py:62-67):
#   (sampleinfo, kmers, base_means, base_stds, base_signal_lens, cent_signals, labels)
FeaturesBatch = Tuple[list, list, list, list, list, list, list]

KILL = "kill"     # sentinel the reader appends and the writer stops on (call_modifications.py:90,291)


def _parse_line(line: str):
    words = line.strip().split("\t")
    return (words[4], "\t".join(words[0:6]), [base2code_dna[x] for x in words[6]],
            [float(x) for x in words[7].split(",")], [float(x) for x in words[8].split(",")],
            [int(x) for x in words[9].split(",")], [float(x) for x in words[10].split(",")], int(words[11]))


def iter_features_batches(features_file: str, f5_batch_num: int = 20) -> Iterator[FeaturesBatch]:
    """Yield queue items: consecutive rows grouped by read id, one item every f5_batch_num reads.

    Same grouping rule as the reference reader (call_modifications.py:35-91): rows of one read are
    never split across items, so results of each read are written together."""
    cur: List[list] = [[], [], [], [], [], [], []]
    r_num = 0
    readid_pre = None
    with open(features_file, "r") as rf:
        for line in rf:
            if not line.strip():
                continue
            readid, info, kmer, means, stds, lens, sig, label = _parse_line(line)
            if readid_pre is None:
                readid_pre = readid
            elif readid != readid_pre:
                r_num += 1
                readid_pre = readid
                if r_num % f5_batch_num == 0:
                    yield tuple(cur)
                    cur = [[], [], [], [], [], [], []]
            for lst, v in zip(cur, (info, kmer, means, stds, lens, sig, label)):
                lst.append(v)
    if len(cur[0]) > 0:
        yield tuple(cur)


def _read_features_file(features_file, features_batch_q, f5_batch_num=20):
    """Queue-feeding form of the reader (reference signature, call_modifications.py:35)."""
    print("read_features process {} starts".format(os.getpid()))
    for item in iter_features_batches(features_file, f5_batch_num):
        features_batch_q.put(item)
    features_batch_q.put(KILL)
    print("read_features process {} ending".format(os.getpid()))


def _read_features_from_fast5s(fast5s, corrected_group, basecall_subgroup, normalize_method, motif_seqs, methyloc,
                               chrom2len, kmer_len, raw_signals_len, methy_label, positions=None):
    """fast5 files -> ONE queue item (reference call_modifications.py:94-122). Means / stds reach the
    model un-rounded on this route, exactly as in the reference (:115-116)."""
    from .extract_features import _extract_features
    features_list, error = _extract_features(fast5s, corrected_group, basecall_subgroup, normalize_method, motif_seqs,
                                             methyloc, chrom2len, kmer_len, raw_signals_len, methy_label, positions)
    sampleinfo, kmers, base_means, base_stds, base_signal_lens, cent_signals, labels = [], [], [], [], [], [], []
    for (chrom, pos, alignstrand, loc_in_ref, readname, strand, k_mer, signal_means, signal_stds, signal_lens,
         kmer_cent_signals, f_methy_label) in features_list:
        sampleinfo.append("\t".join([chrom, str(pos), alignstrand, str(loc_in_ref), readname, strand]))
        kmers.append([base2code_dna[x] for x in k_mer])
        base_means.append(signal_means)
        base_stds.append(signal_stds)
        base_signal_lens.append(signal_lens)
        cent_signals.append(kmer_cent_signals)
        labels.append(f_methy_label)
    return [(sampleinfo, kmers, base_means, base_stds, base_signal_lens, cent_signals, labels)], error


def _call_mods(features_batch: FeaturesBatch, engine, batch_size: int):
    """One queue item -> output rows. Reference: call_modifications.py:149-194.

    Slices the item into batch_size chunks (last one partial), runs the engine on each, normalises
    the two sigmoid outputs in float32 (p_k / (p_0 + p_1)) and formats
    sampleinfo \\t prob_0 \\t prob_1 \\t called_label \\t kmer."""
    sampleinfo, kmers, base_means, base_stds, base_signal_lens, cent_signals, labels = features_batch
    labels = np.reshape(labels, (len(labels)))
    pred_str: List[str] = []
    accuracys = []
    batch_num = 0
    for i in np.arange(0, len(sampleinfo), batch_size):
        batch_s, batch_e = i, i + batch_size
        b_sampleinfo = sampleinfo[batch_s:batch_e]
        b_kmers = kmers[batch_s:batch_e]
        b_labels = labels[batch_s:batch_e]
        if len(b_sampleinfo) > 0:
            activation_logits, prediction = engine.run(
                b_kmers, base_means[batch_s:batch_e], base_stds[batch_s:batch_e],
                base_signal_lens[batch_s:batch_e], cent_signals[batch_s:batch_e])
            accuracys.append(float(np.mean(np.asarray(b_labels) == np.asarray(prediction))))
            for idx in range(b_labels.shape[0]):
                prob_0, prob_1 = activation_logits[idx][0], activation_logits[idx][1]
                prob_0_norm = prob_0 / (prob_0 + prob_1)
                prob_1_norm = prob_1 / (prob_0 + prob_1)
                pred_str.append("\t".join([b_sampleinfo[idx], str(prob_0_norm), str(prob_1_norm),
                                           str(prediction[idx]),
                                           "".join([code2base_dna[x] for x in b_kmers[idx]])]))
            batch_num += 1
    accuracy = np.mean(accuracys) if accuracys else float("nan")
    return pred_str, accuracy, batch_num


def _call_mods_q(engine, features_batch_q, pred_str_q, batch_size):
    """Consume queue items until the sentinel (reference _call_mods_q, call_modifications.py:197-236)."""
    print("call_mods process {} starts".format(os.getpid()))
    count = 0
    while True:
        features_batch = features_batch_q.get()
        if isinstance(features_batch, str) and features_batch == KILL:
            break
        pred_str, _, batch_num = _call_mods(features_batch, engine, batch_size)
        pred_str_q.put(pred_str)
        count += batch_num
    print("call_mods process {} ending, proceed {} batches".format(os.getpid(), count))


def _write_predstr_to_file(write_fp, predstr_q):
    """Writer (reference call_modifications.py:283-297): one flush per queue item."""
    print("write process {} starts".format(os.getpid()))
    with open(write_fp, "w") as wf:
        while True:
            pred_str = predstr_q.get()
            if isinstance(pred_str, str) and pred_str == KILL:
                break
            for one_pred_str in pred_str:
                wf.write(one_pred_str + "\n")
            wf.flush()
    print("write process {} ending".format(os.getpid()))


def load_model_weights(model_path: str, kmer_len: int, cent_signals_len: int, class_num: int,
                       is_cnn: bool = True, is_rnn: bool = True, is_base: bool = True):
    """`--model_path` resolution. The reference passes a TensorFlow checkpoint prefix to Saver.restore
    (call_modifications.py:210-211; files <prefix>.index / .data-00000-of-00001); such a prefix is imported by
    deepsignal_amd.tf_checkpoint and returned as a weight dict. Anything else is taken as a DSAMDW01 weight file
    and returned as None (the engine reads it itself)."""
    from . import tf_checkpoint
    if tf_checkpoint.is_checkpoint(model_path):
        return tf_checkpoint.checkpoint_to_weights(model_path, kmer_len, cent_signals_len, class_num,
                                                   is_cnn=is_cnn, is_rnn=is_rnn, is_base=is_base)
    if not os.path.exists(model_path):
        raise FileNotFoundError("%s: neither a TensorFlow checkpoint prefix (no %s.index) nor a weight file"
                                % (model_path, model_path))
    return None


def make_engine(model_path: str, kmer_len: int, cent_signals_len: int, class_num: int, batch_size: int,
                is_cnn: bool = True, is_rnn: bool = True, is_base: bool = True, device: int = 0,
                precision: str = "fp32"):
    """Model(...) + Session + Saver.restore (reference call_modifications.py:203-212)."""
    from .engine import Engine
    weights = load_model_weights(model_path, kmer_len, cent_signals_len, class_num, is_cnn, is_rnn, is_base)
    eng = Engine(kmer_len=kmer_len, signal_len=cent_signals_len, class_num=class_num, device=device,
                 max_batch=batch_size, is_cnn=is_cnn, is_rnn=is_rnn, is_base=is_base, precision=precision)
    if weights is None:
        eng.load_weights_file(model_path)
    else:
        eng.load_weights(weights)
    return eng


def _distributed_context(dist):
    """(dist module or None, rank, world, local device). `dist` = an initialised torch.distributed, or None to
    look at the launcher's environment (torchrun / torch.distributed.run sets WORLD_SIZE, RANK, LOCAL_RANK)."""
    if dist is not None:
        return dist, dist.get_rank(), dist.get_world_size(), int(os.environ.get("LOCAL_RANK", dist.get_rank()))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None, 0, 1, 0
    import torch.distributed as tdist
    local = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    if not tdist.is_initialized():
        import torch
        if torch.cuda.device_count() > 0:
            torch.cuda.set_device(local)
            tdist.init_process_group("nccl")           # RCCL over xGMI
        else:
            tdist.init_process_group("gloo")
    return tdist, tdist.get_rank(), tdist.get_world_size(), local


def _run_item(engine, item, batch_size):
    """All rows of one queue item through the engine in batch_size chunks; the chunks are kept in flight through the
    asynchronous boundary when the engine has one."""
    n = len(item.labels)
    spans = [(s, min(n, s + batch_size)) for s in range(0, n, batch_size)]
    args = lambda s, e: (item.kmer[s:e], item.means[s:e], item.stds[s:e], item.lens[s:e], item.signals[s:e])
    acts, preds = [], []
    if hasattr(engine, "submit") and hasattr(engine, "wait") and batch_size <= getattr(engine, "max_batch", 0):
        pending = []
        for s, e in spans:
            if len(pending) == engine.slots:
                ai, pi = engine.wait(pending.pop(0))
                acts.append(ai); preds.append(pi)
            pending.append(engine.submit(*args(s, e)))
        for t in pending:
            ai, pi = engine.wait(t)
            acts.append(ai); preds.append(pi)
    else:
        for s, e in spans:
            ai, pi = engine.run(*args(s, e))
            acts.append(ai); preds.append(pi)
    return np.concatenate(acts), np.concatenate(preds)


def _call_mods_sharded(reader_items, engine, batch_size, result_file, dist, rank, world, device=None):
    """One process per GPU: queue items hold whole reads, item k is run by rank k % world on its own
    weight replica; after every round of `world` items the 12 B/site results are gathered to rank 0 (the only
    collective), which formats and writes the rows in file order -- reads stay contiguous (README.rst:15)."""
    from . import fastio, sharding
    nsites = 0
    wf = open(result_file, "wb") if rank == 0 else None
    round_items = []

    def flush():
        nonlocal nsites
        if not round_items:
            return
        base = 0
        act = np.zeros((0, engine.class_num), np.float32)
        pred = np.zeros((0,), np.int32)
        index = np.zeros((0,), np.int64)
        for k, item, a, p in round_items:
            n = len(item.labels)
            if a is not None:
                act, pred, index = a, p, np.arange(base, base + n, dtype=np.int64)
            base += n
        g_act, g_pred = sharding.gather_results(act, pred, index, dist, dst=0, device=device)
        if rank == 0:
            base = 0
            for k, item, _, _ in round_items:
                n = len(item.labels)
                wf.write(fastio.format_rows(item.info, item.info_off, g_act[base:base + n], g_pred[base:base + n],
                                            item.kmer))
                base += n
            wf.flush()
        nsites += base
        round_items.clear()

    for k, item in enumerate(reader_items):
        a = p = None
        if k % world == rank:
            a, p = _run_item(engine, item, batch_size)
        round_items.append((k, item, a, p))
        if len(round_items) == world:
            flush()
    flush()
    if wf is not None:
        wf.close()
    return nsites


def call_mods(input_path, model_path, result_file, kmer_len, cent_signals_len,
              batch_size, learning_rate, class_num, nproc, is_gpu, is_rnn, is_base, is_cnn,
              f5_args, engine=None, f5_batch_num=None, native_io=True, precision="fp32", dist=None):
    """Feature-file mode of the reference's call_mods (call_modifications.py:417-495).

    learning_rate / nproc / is_gpu are accepted for signature compatibility: inference ignores the
    learning rate and the engine always runs on the GPU. `f5_args` is the reference's
    tuple (its first element is f5_batch_num); a directory as input_path takes the fast5 route
    (features extracted on the host, deepsignal_amd/extract_features.py). With the native reader the engine is
    driven through its asynchronous boundary (submit / wait, several batches in flight) and, when launched by
    torch.distributed.run with WORLD_SIZE > 1, one process per GPU shares the reads."""
    start = time.time()
    if os.path.isdir(input_path):
        return _call_mods_from_fast5s(input_path, model_path, result_file, kmer_len, cent_signals_len, batch_size,
                                      class_num, is_rnn, is_base, is_cnn, f5_args, engine, nproc=nproc)
    if f5_batch_num is None:
        f5_batch_num = f5_args[0] if f5_args else 50
    dist, rank, world, local = _distributed_context(dist)
    own = engine is None
    if own:
        engine = make_engine(model_path, kmer_len, cent_signals_len, class_num, batch_size,
                             is_cnn=is_cnn, is_rnn=is_rnn, is_base=is_base, device=local, precision=precision)
    nsites = 0
    if world > 1:
        # launched as `python -m torch.distributed.run --nproc-per-node N -m deepsignal_amd.deepsignal call_mods ...`
        if not native_io:
            raise ValueError("multi-GPU call_mods uses the native feature reader (native_io=True)")
        from . import fastio
        reader = fastio.FeatureReader(input_path, kmer_len, cent_signals_len)
        device = None
        if dist.get_backend() == "nccl":
            import torch
            device = torch.device("cuda", local)
        nsites = _call_mods_sharded(reader.items(f5_batch_num), engine, batch_size, result_file, dist, rank, world, device)
        reader.close()
    elif native_io:
        # row f1: native reader (host threads) + native row formatter; same items, same row text
        from . import fastio
        reader = fastio.FeatureReader(input_path, kmer_len, cent_signals_len)
        # parse the next item on a helper thread while the engine works on the current one
        # (the native calls release the GIL); a bounded queue keeps memory flat
        import queue
        import threading
        q = queue.Queue(maxsize=3)

        def _produce():
            try:
                for it in reader.items(f5_batch_num):
                    q.put(it)
                q.put(None)
            except BaseException as exc:          # surfaced on the consumer side
                q.put(exc)

        th = threading.Thread(target=_produce, daemon=True)
        th.start()
        import collections
        inflight = collections.deque()
        pipelined = hasattr(engine, "submit") and hasattr(engine, "wait") and batch_size <= getattr(engine, "max_batch", 0)
        with open(result_file, "wb") as wf:
            def drain(limit):
                while len(inflight) > limit:
                    ticket, segs_ = inflight.popleft()
                    act_, pred_ = engine.wait(ticket)
                    o = 0
                    for it_, s_, e_ in segs_:
                        m_ = e_ - s_
                        wf.write(fastio.format_rows(it_.info, it_.info_off[s_:e_ + 1], act_[o:o + m_], pred_[o:o + m_],
                                                    it_.kmer[s_:e_]))
                        o += m_

            def submit(segs_):
                # asynchronous boundary: up to `slots` batches in flight; rows leave in submission order
                drain(engine.slots - 1)
                if len(segs_) == 1:
                    it_, s_, e_ = segs_[0]
                    arrs = (it_.kmer[s_:e_], it_.means[s_:e_], it_.stds[s_:e_], it_.lens[s_:e_], it_.signals[s_:e_])
                else:
                    arrs = tuple(np.concatenate([getattr(it_, k)[s_:e_] for it_, s_, e_ in segs_])
                                 for k in ("kmer", "means", "stds", "lens", "signals"))
                inflight.append((engine.submit(*arrs), segs_))

            # Pipelined route: batches are filled ACROSS queue items (a site's result does not depend on its batch
            # mates), so the engine sees full batch_size forwards instead of one ragged tail per item; rows are still
            # written in file order, a read's rows together.
            segs, count = [], 0
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                n = len(item.labels)
                if pipelined:
                    s = 0
                    while s < n:
                        take = min(n - s, batch_size - count)
                        segs.append((item, s, s + take))
                        count += take
                        s += take
                        if count == batch_size:
                            submit(segs)
                            segs, count = [], 0
                else:
                    for s in range(0, n, batch_size):
                        e = min(n, s + batch_size)
                        act, pred = engine.run(item.kmer[s:e], item.means[s:e], item.stds[s:e], item.lens[s:e],
                                               item.signals[s:e])
                        wf.write(fastio.format_rows(item.info, item.info_off[s:e + 1], act, pred, item.kmer[s:e]))
                nsites += n
            if count:
                submit(segs)
            drain(0)
            wf.flush()
        th.join()
        reader.close()
    else:
        with open(result_file, "w") as wf:
            for item in iter_features_batches(input_path, f5_batch_num):
                pred_str, _, _ = _call_mods(item, engine, batch_size)
                for row in pred_str:
                    wf.write(row + "\n")
                wf.flush()
                nsites += len(pred_str)
    if own:
        engine.close()
    if rank == 0:
        print("call_mods costs %.2f seconds.. (%d sites)" % (time.time() - start, nsites))
    return nsites


def _fast5_task(task):
    """Worker of the fast5 route: one batch of files -> one queue item (picklable wrapper)."""
    return _read_features_from_fast5s(*task)


def _call_mods_from_fast5s(fast5_dir, model_path, result_file, kmer_len, cent_signals_len, batch_size, class_num,
                           is_rnn, is_base, is_cnn, f5_args, engine=None, nproc=1):
    """fast5-directory mode (reference call_modifications.py:431-448 + :300-414, single process):
    batches of f5_batch_num files -> features on the host -> engine -> rows. Needs h5py for the HDF5 files."""
    from . import extract_features as ef
    (f5_batch_num, is_recursive, corrected_group, basecall_subgroup, is_dna, normalize_method, motifs, mod_loc,
     methy_label, position_file, reference_path) = f5_args
    start = time.time()
    fast5s = ef.get_fast5s(fast5_dir, is_recursive)
    print("{} fast5 files in total..".format(len(fast5s)))
    motif_seqs = ef.get_motif_seqs(motifs, is_dna)
    chrom2len = ef.read_reference_lengths(reference_path)       # contig lengths (reference utils/ref_reader.py:7-13)
    positions = ef.read_position_file(position_file)
    own = engine is None
    if own:
        engine = make_engine(model_path, kmer_len, cent_signals_len, class_num, batch_size,
                             is_cnn=is_cnn, is_rnn=is_rnn, is_base=is_base)
    errors = nsites = 0
    tasks = [(fast5s[i:i + f5_batch_num], corrected_group, basecall_subgroup, normalize_method, motif_seqs, mod_loc,
              chrom2len, kmer_len, cent_signals_len, methy_label, positions) for i in range(0, len(fast5s), f5_batch_num)]
    pool = None
    if nproc > 2 and len(tasks) > 1:
        # the reference runs nproc - 1 extraction processes next to the GPU process (call_modifications.py:431-448);
        # here nproc - 1 workers extract file batches (in order) while this process drives the engine
        import multiprocessing as mp
        pool = mp.get_context("spawn").Pool(min(nproc - 1, len(tasks)))
        results = pool.imap(_fast5_task, tasks)
    else:
        results = (_fast5_task(t) for t in tasks)
    with open(result_file, "w") as wf:
        for batches, err in results:
            errors += err
            for fb in batches:
                pred_str, _, _ = _call_mods(fb, engine, batch_size)
                for row in pred_str:
                    wf.write(row + "\n")
                nsites += len(pred_str)
            wf.flush()
    if pool is not None:
        pool.close()
        pool.join()
    if own:
        engine.close()
    print("%d of %d fast5 files failed.." % (errors, len(fast5s)))
    print("call_mods costs %.2f seconds.. (%d sites)" % (time.time() - start, nsites))
    return nsites
