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
from typing import Iterator, List, NamedTuple, Optional, Sequence, Tuple

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


COLLECTIVE_TIMEOUT_S = 600.0      # process-group timeout of a multi-GPU call_mods (see _distributed_context)
# sites per forward the engine is created for (see make_engine). TSV -> TSV on MI355X, 327,680 rows (tools/e2e_profile.py): fp32
# 505 / 504 / 526 / 532 k sites/s at 512 / 1024 / 2048 / 4096 (fewer, larger H2D copies and Python round trips per site; the
# resident-input rate itself gains 2 %); bf16_all 1.3 M at 512 against 1.9 M at 4096
ENGINE_BATCH = {"fp32": 4096, "bf16": 4096, "bf16_all": 4096, "bf16x3": 4096}


def engine_batch_for(batch_size: int, precision: str, engine_batch: int = 0) -> int:
    """Sites per forward the engine is created for. Default: the larger of --batch_size and ENGINE_BATCH[precision]; an explicit
    `engine_batch` (> 0: the CLI's --engine_batch, or the DS_ENGINE_BATCH environment variable read here, in the Python harness)
    is taken as given -- the escape hatch for a small or shared GPU: device and pinned memory grow with it (about 1 GB per 512
    sites over the engine's slots)."""
    if engine_batch <= 0:
        try:
            engine_batch = int(os.environ.get("DS_ENGINE_BATCH", "0"))
        except ValueError:
            raise ValueError("DS_ENGINE_BATCH must be an integer")
    if engine_batch > 0:
        return engine_batch
    return max(batch_size, ENGINE_BATCH.get(precision, batch_size))


def make_engine(model_path: str, kmer_len: int, cent_signals_len: int, class_num: int, batch_size: int,
                is_cnn: bool = True, is_rnn: bool = True, is_base: bool = True, device: int = 0,
                precision: str = "fp32", engine_batch: int = 0):
    """Model(...) + Session + Saver.restore (reference call_modifications.py:203-212)."""
    from .engine import Engine
    weights = load_model_weights(model_path, kmer_len, cent_signals_len, class_num, is_cnn, is_rnn, is_base)
    # `batch_size` is the reference's rows-per-sess.run (call_modifications.py:157-166); a site's result does not depend on
    # its batch mates (tests: the same bits alone, in a sub-batch, in a full batch), so the engine is sized for the batch the
    # GPU wants -- ENGINE_BATCH -- and the row pipeline fills THAT (the bf16 modes need 4096 sites for their 3.3 M sites/s, 2.6 M at
    # 512; fp32 end to end gains 5 % from the larger copies)
    want = engine_batch_for(batch_size, precision, engine_batch)
    kw = dict(kmer_len=kmer_len, signal_len=cent_signals_len, class_num=class_num, device=device, is_cnn=is_cnn, is_rnn=is_rnn,
              is_base=is_base, precision=precision)
    try:
        eng = Engine(max_batch=want, **kw)
    except RuntimeError as exc:
        # the GPU has no room for the large batch (another job's memory, a smaller device): the user's own batch size still works
        if "hipMalloc" not in str(exc) or want <= batch_size:
            raise
        print("note: no device memory for an engine of %d sites per forward (%s); falling back to --batch_size %d"
              % (want, exc, batch_size), file=sys.stderr)
        eng = Engine(max_batch=batch_size, **kw)
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
        import datetime
        import torch
        # an explicit timeout on every collective of the job: a peer that never joins (crashed rank, wrong WORLD_SIZE) ends
        # the run with an error -- RCCL's watchdog aborts the process, gloo raises -- instead of hanging the row gather
        timeout = datetime.timedelta(seconds=COLLECTIVE_TIMEOUT_S)
        ngpu = torch.cuda.device_count()
        # DS_DIST_BACKEND=gloo (read by this Python harness only, never by the library): the collectives on host tensors and
        # the ranks dealt round-robin over the GPUs that exist -- how the launcher route is tested with two ranks on the one
        # GPU of a test box; the product default is one rank per GPU on RCCL
        backend = os.environ.get("DS_DIST_BACKEND") or ("nccl" if ngpu > 0 else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            tdist.init_process_group("nccl", timeout=timeout)           # RCCL over xGMI
        else:
            if ngpu > 0:
                local = local % ngpu
            tdist.init_process_group(backend, timeout=timeout)
    return tdist, tdist.get_rank(), tdist.get_world_size(), local


class _RowPipeline:
    """Rows of queue items through the engine, results formatted and handed to `sink(tag, row_bytes)` in feed order.

    With an engine that has the asynchronous boundary (submit / wait) batches are filled ACROSS queue items (a site's
    result does not depend on its batch mates), so the engine sees full batch_size forwards instead of one ragged tail
    per item, and up to `engine.slots` batches are in flight while the caller parses the next item. A batch that
    straddles items is handed over as row segments (`submit_parts`: gathered by the library straight into its pinned
    staging buffer, no concatenated copy), and finished batches are formatted and sunk by ONE helper thread in
    completion = feed order (the native formatter releases the GIL), so the feeding thread only submits and waits.
    Rows still leave in file order, a read's rows together. `tag` travels with every row (the sharded path uses it to
    know which work unit a row belongs to)."""

    def __init__(self, engine, batch_size, sink):
        import collections
        import queue
        import threading
        from . import fastio
        self.engine, self.batch_size, self.sink, self.fastio = engine, batch_size, sink, fastio
        self.pipelined = hasattr(engine, "submit") and hasattr(engine, "wait") and \
            batch_size <= getattr(engine, "max_batch", 0)
        if self.pipelined:
            self.batch_size = batch_size = int(engine.max_batch)      # fill whole engine batches (see make_engine)
        self.inflight = collections.deque()
        self.segs, self.count = [], 0
        self.nsites = 0
        self._outq = self._worker = self._error = None
        self._queued, self._qlock = {}, threading.Lock()      # tag -> batches waited but not yet through the sink
        if self.pipelined:
            self._outq = queue.Queue(maxsize=4 * max(1, getattr(engine, "slots", 1)))
            self._worker = threading.Thread(target=self._format_loop, daemon=True)
            self._worker.start()

    def _emit(self, seg, act, pred):
        tag, it, s, e = seg
        self.sink(tag, self.fastio.format_rows(it.info, it.info_off[s:e + 1], act, pred, it.kmer[s:e]))

    def _format_loop(self):
        while True:
            job = self._outq.get()
            try:
                if job is None:
                    return
                segs, act, pred = job
                try:
                    if self._error is None:
                        o = 0
                        for seg in segs:
                            m = seg[3] - seg[2]
                            self._emit(seg, act[o:o + m], pred[o:o + m])
                            o += m
                finally:
                    with self._qlock:
                        for seg in segs:
                            self._queued[seg[0]] -= 1
                            if not self._queued[seg[0]]:
                                del self._queued[seg[0]]
            except BaseException as exc:      # surfaced by the feeding thread (flush / next drain)
                self._error = exc
            finally:
                self._outq.task_done()

    def _check_worker(self):
        if self._error is not None:
            exc, self._error = self._error, None
            raise exc

    def _drain(self, limit):
        while len(self.inflight) > limit:
            ticket, segs = self.inflight.popleft()
            act, pred = self.engine.wait(ticket)
            self._check_worker()
            with self._qlock:
                for seg in segs:
                    self._queued[seg[0]] = self._queued.get(seg[0], 0) + 1
            self._outq.put((segs, act, pred))

    def live_tags(self):
        """Tags with rows still inside the pipeline (being filled, in flight, or waiting for the formatter)."""
        with self._qlock:
            live = set(self._queued)
        live.update(seg[0] for seg in self.segs)
        live.update(seg[0] for _, segs in self.inflight for seg in segs)
        return live

    def _submit(self):
        segs, self.segs, self.count = self.segs, [], 0
        self._drain(self.engine.slots - 1)
        parts = [(it.kmer[s:e], it.means[s:e], it.stds[s:e], it.lens[s:e], it.signals[s:e]) for _, it, s, e in segs]
        if len(parts) == 1:
            ticket = self.engine.submit(*parts[0])
        elif hasattr(self.engine, "submit_parts"):
            ticket = self.engine.submit_parts(parts)
        else:
            ticket = self.engine.submit(*(np.concatenate([p[j] for p in parts]) for j in range(5)))
        self.inflight.append((ticket, segs))

    def feed(self, item, tag=None):
        n = len(item.labels)
        self.nsites += n
        if not self.pipelined:
            for s in range(0, n, self.batch_size):
                e = min(n, s + self.batch_size)
                act, pred = self.engine.run(item.kmer[s:e], item.means[s:e], item.stds[s:e], item.lens[s:e],
                                            item.signals[s:e])
                self._emit((tag, item, s, e), act, pred)
            return
        s = 0
        while s < n:
            take = min(n - s, self.batch_size - self.count)
            self.segs.append((tag, item, s, s + take))
            self.count += take
            s += take
            if self.count == self.batch_size:
                self._submit()

    def flush(self):
        """Everything fed so far has reached the sink when this returns."""
        if self.pipelined:
            if self.count:
                self._submit()
            self._drain(0)
            self._outq.join()
            self._check_worker()

    def close(self):
        """Stop the helper thread (after a flush; idempotent)."""
        if self._worker is not None:
            self._outq.put(None)
            self._worker.join()
            self._worker = None


def _prefetch(iterable, depth=3):
    """Run `iterable` on a helper thread (the native parser releases the GIL), a bounded queue ahead of the consumer.
    The producer has always EXITED when this generator is finished or closed -- also when the consumer stops early
    (an engine or formatter error unwinds through the `for` loop and closes the generator): a stop flag is set, the
    queue is drained so a blocked put() returns, and the thread is joined. Only then may the caller close what the
    iterable reads from (the native reader unmaps its file and deletes its thread team)."""
    import queue
    import threading
    q = queue.Queue(maxsize=depth)
    done = object()
    stop = threading.Event()

    def put(x):
        while not stop.is_set():
            try:
                q.put(x, timeout=0.05)
                return True
            except queue.Full:
                pass
        return False

    def produce():
        try:
            for x in iterable:
                if not put(x):
                    return
            put(done)
        except BaseException as exc:          # surfaced on the consumer side
            put(exc)

    th = threading.Thread(target=produce, daemon=True)
    th.start()
    try:
        while True:
            x = q.get()
            if x is done:
                break
            if isinstance(x, BaseException):
                raise x
            yield x
    finally:
        stop.set()
        while th.is_alive():
            try:
                q.get_nowait()
            except queue.Empty:
                pass
            th.join(timeout=0.05)


# Sharded feature-file mode: the file is cut into byte ranges of whole reads of about this many bytes; range k belongs
# to rank k % world (about 9 k sites of 3.5 kB text per 32 MB: tens of milliseconds of GPU work per range)
SHARD_CHUNK_BYTES = 32 << 20


def _local_world(world):
    return max(1, min(world, int(os.environ.get("LOCAL_WORLD_SIZE", world))))


def _call_mods_sharded(input_path, engine, batch_size, result_file, kmer_len, cent_signals_len, f5_batch_num,
                       dist, rank, world, device=None):
    """One process per GPU. Each rank parses ONLY its own byte ranges of the feature file (cut at read
    boundaries by ds_tsv_align, which every rank computes alike -- per-rank parsing cost is 1/world of the file),
    runs them on its own weight replica and formats its own rows; the rows go to rank 0 through
    sharding.OrderedRowGather, whose communication thread keeps the collectives off this thread, and rank 0 writes
    them in file order -- reads stay contiguous (README.rst:15). Returns the job-wide site count on every rank."""
    from . import fastio, sharding
    threads = max(1, fastio.usable_cpus() // _local_world(world))
    reader = fastio.FeatureReader(input_path, kmer_len, cent_signals_len, nthreads=threads)
    per_rank = max(1, min(4096, reader.size // max(1, world * SHARD_CHUNK_BYTES)))
    nchunks = world * per_rank
    cuts = reader.cut_points(nchunks)
    gather = sharding.OrderedRowGather(dist, rank, world, result_file, nrounds=per_rank, device=device)
    mine = list(range(rank, nchunks, world))
    bufs = {c: [] for c in mine}
    fed_all = set()                        # chunks whose last row has been fed to the pipeline
    next_out = [0]                         # index into `mine` of the next chunk to hand over

    def sink(tag, data):
        bufs[tag].append(data)

    pipe = _RowPipeline(engine, batch_size, sink)
    # rows of chunk c are complete once everything fed up to the chunk's end has drained; with batches filled across
    # chunk borders that is known when a LATER batch drains, so hand chunks over lazily and flush at the very end

    def hand_over(upto_complete):
        while next_out[0] < len(mine) and mine[next_out[0]] in upto_complete:
            c = mine[next_out[0]]
            gather.put(b"".join(bufs.pop(c)))
            upto_complete.discard(c)
            next_out[0] += 1

    def rows():
        for c in mine:
            reader.set_range(cuts[c], cuts[c + 1])
            for item in reader.items(f5_batch_num):
                yield c, item
            yield c, None

    failed = False
    items = _prefetch(rows())
    try:
        pending_done = []                  # chunks fully fed, waiting for their rows to drain
        for c, item in items:
            if item is not None:
                pipe.feed(item, c)
                continue
            pending_done.append(c)
            # a chunk's rows have all drained when nothing of it is buffered in the pipeline any more
            live = pipe.live_tags()
            fed_all.update(x for x in pending_done if x not in live)
            pending_done = [x for x in pending_done if x in live]
            hand_over(fed_all)
        pipe.flush()
        fed_all.update(pending_done)
        hand_over(fed_all)
    except BaseException:
        failed = True
        raise
    finally:
        try:
            total, _ = gather.close(pipe.nsites, 0, failed=failed)
        finally:
            items.close()                  # the parser thread has exited (never inside ds_tsv_next when the reader closes)
            pipe.close()
            reader.close()
    return total


def call_mods(input_path, model_path, result_file, kmer_len, cent_signals_len,
              batch_size, learning_rate, class_num, nproc, is_gpu, is_rnn, is_base, is_cnn,
              f5_args, engine=None, f5_batch_num=None, native_io=True, precision="fp32", dist=None, force_sharded=False,
              engine_batch=0):
    """The reference's call_mods (call_modifications.py:417-495), same signature and argument meaning.

    learning_rate / is_gpu are accepted for signature compatibility: inference ignores the learning rate
     and the engine always runs on the GPU. `f5_args` is the reference's 11-tuple
    (call_modifications.py:428-429): (is_recursive, corrected_group, basecall_subgroup, reference_path, is_dna,
    normalize_method, motifs, mod_loc, methy_label, f5_batch_num, position_file); the keyword `f5_batch_num`
    overrides its entry. A directory as input_path takes the fast5 route (features extracted on the host,
    deepsignal_amd/extract_features.py). With the native reader the engine is driven through its asynchronous boundary
    (submit / wait, several batches in flight) and, when launched by torch.distributed.run with WORLD_SIZE > 1, one
    process per GPU takes its share of the reads. force_sharded=True takes that multi-process route (byte ranges, row
    gather through the process group's collectives) even in a world of one: one rank under the launcher runs the same
    RCCL calls as eight."""
    start = time.time()
    f5 = _unpack_f5_args(f5_args, f5_batch_num)
    dist, rank, world, local = _distributed_context(dist)
    if precision in ("bf16", "bf16_all") and rank == 0 and engine is None:
        # measured, DESIGN.md section 9: harmless on random-init weights, NOT on weights at a trained model's scale
        print("note: --precision %s stores activations as bf16: on trained-scale weights expect probabilities to move by ~0.02 (up "
              "to ~0.2) and a few per cent of the calls nearest the threshold to change label. --precision bf16x3 is the fast mode "
              "with fp32-class results (held to the fp32 parity bars, ~1.4-1.5x the fp32 rate); fp32 is the reference numerics"
              % precision, file=sys.stderr)
    device = None
    if dist is not None and (world > 1 or force_sharded) and dist.get_backend() == "nccl":
        import torch
        device = torch.device("cuda", local)
    own = engine is None
    if own:
        engine = make_engine(model_path, kmer_len, cent_signals_len, class_num, batch_size,
                             is_cnn=is_cnn, is_rnn=is_rnn, is_base=is_base, device=local, precision=precision,
                             engine_batch=engine_batch)
    try:
        if os.path.isdir(input_path):
            nsites = _call_mods_from_fast5s(input_path, result_file, kmer_len, cent_signals_len, batch_size, f5, engine,
                                            nproc=nproc, dist=dist, rank=rank, world=world, device=device)
        elif world > 1 or (force_sharded and dist is not None):
            # launched as `python -m torch.distributed.run --nproc-per-node N -m deepsignal_amd.deepsignal call_mods ...`
            if not native_io:
                raise ValueError("multi-GPU call_mods uses the native feature reader (native_io=True)")
            nsites = _call_mods_sharded(input_path, engine, batch_size, result_file, kmer_len, cent_signals_len,
                                        f5.f5_batch_num, dist, rank, world, device)
        elif native_io:
            # row f1: native reader (host threads, one item ahead on a helper thread) + native row formatter;
            # same items, same row text
            from . import fastio
            reader = fastio.FeatureReader(input_path, kmer_len, cent_signals_len)
            try:
                with open(result_file, "wb") as wf:
                    pipe = _RowPipeline(engine, batch_size, lambda tag, data: wf.write(data))
                    items = _prefetch(reader.items(f5.f5_batch_num))
                    try:
                        for item in items:
                            pipe.feed(item)
                        pipe.flush()
                    finally:
                        items.close()          # the parser thread has exited before the reader goes away
                        pipe.close()
                    wf.flush()
                nsites = pipe.nsites
            finally:
                reader.close()
        else:
            nsites = 0
            with open(result_file, "w") as wf:
                for item in iter_features_batches(input_path, f5.f5_batch_num):
                    pred_str, _, _ = _call_mods(item, engine, batch_size)
                    for row in pred_str:
                        wf.write(row + "\n")
                    wf.flush()
                    nsites += len(pred_str)
    finally:
        if own:
            engine.close()
    if rank == 0:
        print("call_mods costs %.2f seconds.. (%d sites)" % (time.time() - start, nsites))
    return nsites


class _F5Args(NamedTuple):
    """The reference's f5_args, by name (deepsignal.py:83-84, call_modifications.py:428-429)."""
    is_recursive: bool = True
    corrected_group: str = "RawGenomeCorrected_000"
    basecall_subgroup: str = "BaseCalled_template"
    reference_path: Optional[str] = None
    is_dna: bool = True
    normalize_method: str = "mad"
    motifs: str = "CG"
    mod_loc: int = 0
    methy_label: int = 1
    f5_batch_num: int = 50
    position_file: Optional[str] = None


def _unpack_f5_args(f5_args, f5_batch_num=None) -> "_F5Args":
    """f5_args in the reference's order; None / () take the CLI defaults. Anything else is an error: a tuple in another
    layout would silently mis-assign every field."""
    if f5_args is None or len(f5_args) == 0:
        f5 = _F5Args()
    elif len(f5_args) == len(_F5Args._fields):
        f5 = _F5Args(*f5_args)
    else:
        raise ValueError("f5_args must be the reference's 11-tuple %s" % (_F5Args._fields,))
    if not isinstance(f5.f5_batch_num, (int, np.integer)) or isinstance(f5.f5_batch_num, bool) or f5.f5_batch_num < 1:
        raise ValueError("f5_args[9] (f5_batch_num) must be a positive int, got %r -- is the tuple in the reference's "
                         "order %s?" % (f5.f5_batch_num, _F5Args._fields))
    if f5_batch_num is not None:
        f5 = f5._replace(f5_batch_num=int(f5_batch_num))
    return f5


def _fast5_task(task):
    """Worker of the fast5 route: one batch of files -> one queue item (picklable wrapper)."""
    return _read_features_from_fast5s(*task)


def _call_mods_from_fast5s(fast5_dir, result_file, kmer_len, cent_signals_len, batch_size, f5, engine, nproc=1,
                           dist=None, rank=0, world=1, device=None):
    """fast5-directory mode (reference call_modifications.py:431-448 + :300-414): batches of f5_batch_num files ->
    features on the host -> engine -> rows. Needs h5py for the HDF5 files. With world > 1 (one process per GPU) file
    batch k belongs to rank k % world -- extraction, the forward and row formatting all happen on the owning rank -- and
    rank 0 writes the batches' rows in batch order through sharding.OrderedRowGather."""
    from . import extract_features as ef
    fast5s = ef.get_fast5s(fast5_dir, f5.is_recursive)
    if rank == 0:
        print("{} fast5 files in total..".format(len(fast5s)))
    motif_seqs = ef.get_motif_seqs(f5.motifs, f5.is_dna)
    chrom2len = ef.read_reference_lengths(f5.reference_path)    # contig lengths (reference utils/ref_reader.py:7-13)
    positions = ef.read_position_file(f5.position_file)
    errors = nsites = 0
    tasks = [(fast5s[i:i + f5.f5_batch_num], f5.corrected_group, f5.basecall_subgroup, f5.normalize_method, motif_seqs,
              f5.mod_loc, chrom2len, kmer_len, cent_signals_len, f5.methy_label, positions)
             for i in range(0, len(fast5s), f5.f5_batch_num)]
    mine = tasks[rank::world]
    pool = None
    if nproc > 2 and len(mine) > 1:
        # the reference runs nproc - 1 extraction processes next to the GPU process (call_modifications.py:431-448);
        # here nproc - 1 workers extract file batches (in order) while this process drives the engine
        import multiprocessing as mp
        pool = mp.get_context("spawn").Pool(min(nproc - 1, len(mine)))
        results = pool.imap(_fast5_task, mine)
    else:
        results = (_fast5_task(t) for t in mine)
    gather = None
    failed = False
    if world > 1:
        from . import sharding
        gather = sharding.OrderedRowGather(dist, rank, world, result_file, nrounds=(len(tasks) + world - 1) // world,
                                           device=device)
    wf = open(result_file, "w") if gather is None else None
    try:
        for batches, err in results:
            errors += err
            rows = []
            for fb in batches:
                pred_str, _, _ = _call_mods(fb, engine, batch_size)
                rows.extend(pred_str)
            nsites += len(rows)
            text = "".join(r + "\n" for r in rows)
            if gather is None:
                wf.write(text)
                wf.flush()
            else:
                gather.put(text.encode())
    except BaseException:
        failed = True
        raise
    finally:
        if pool is not None:
            pool.close()
            pool.join()
        if wf is not None:
            wf.close()
        if gather is not None:
            nsites, errors = gather.close(nsites, errors, failed=failed)
    if rank == 0:
        print("%d of %d fast5 files failed.." % (errors, len(fast5s)))
    return nsites
