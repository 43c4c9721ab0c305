"""Multi-GPU sharding of call_mods: sites are partitioned BY READ (all sites of a read stay on one
rank, so a read's results stay contiguous — the reference's "results of each read be written
together" guarantee, README.rst:15 / call_modifications.py:62-67), every rank holds a full weight
replica, and the only collective is the gather of the 12 B/site results to the writer rank.

Works with any torch.distributed backend: "nccl" (= RCCL over xGMI on MI355X) or "gloo" (CPU tests).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np


def assign_reads(read_ids: Sequence[str], world: int) -> np.ndarray:
    """Rank of every row: distinct reads (in order of first appearance) are dealt round-robin."""
    owner = {}
    ranks = np.empty(len(read_ids), dtype=np.int64)
    for i, rid in enumerate(read_ids):
        if rid not in owner:
            owner[rid] = len(owner) % world
        ranks[i] = owner[rid]
    return ranks


def shard_indices(read_ids: Sequence[str], world: int, rank: int) -> np.ndarray:
    return np.nonzero(assign_reads(read_ids, world) == rank)[0]


def gather_results(act, pred, index, dist=None, dst: int = 0, device=None, as_numpy: bool = True, index_of_rank=None,
                   force_collective: bool = False):
    """Gather (act float32[n_i,C], pred int32[n_i]) from all ranks to `dst`, re-ordered so that row j is the site with
    global index j; returns (act, pred) on dst and (None, None) elsewhere.

    On the wire: act as float32 and pred as int32 -- **12 B/site at C = 2**  -- in two ragged gathers
    (every rank pads to the longest shard). The global indices travel as a third int64 gather only when the writer
    cannot derive them: with `index_of_rank(rank, count) -> int64 tensor` (the sharding rule, e.g. "rank r owns reads
    r, r + world, ...") nothing but the 12 B/site moves. `dist` = torch.distributed (already initialised) or None for
    single-process. as_numpy=False keeps the re-ordered results as tensors on `device` (no device-to-host copy inside the
    call: what bench.py times is the RCCL exchange itself). A world of one takes a shortcut without any collective unless
    force_collective is set (one rank under the launcher: the same RCCL calls as N > 1, on one GPU)."""
    import torch
    act_t = torch.as_tensor(act, dtype=torch.float32)
    pred_t = torch.as_tensor(pred, dtype=torch.int32)
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force_collective):
        idx_t = torch.as_tensor(index, dtype=torch.int64) if index is not None else index_of_rank(0, act_t.shape[0])
        order = torch.argsort(idx_t.cpu())
        return act_t.cpu()[order].numpy(), pred_t.cpu()[order].numpy()
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else act_t.device
    C = act_t.shape[1]
    n = act_t.shape[0]
    n_local = torch.tensor([n], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    counts = [int(c.item()) for c in counts]
    nmax = max(max(counts), 1)

    def padded(t, dtype, width=None):
        shape = (nmax,) if width is None else (nmax, width)
        buf = torch.zeros(shape, dtype=dtype, device=dev)
        if n:
            buf[:n] = t.to(dev, dtype)
        return buf

    def gather(buf):
        gl = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
        dist.gather(buf, gl, dst=dst)
        return gl

    g_act = gather(padded(act_t, torch.float32, C))
    g_pred = gather(padded(pred_t, torch.int32))
    g_idx = None
    if index_of_rank is None:
        g_idx = gather(padded(torch.as_tensor(index, dtype=torch.int64), torch.int64))
    if rank != dst:
        return None, None
    total = sum(counts)
    t_act = torch.empty((total, C), dtype=torch.float32, device=dev)
    t_pred = torch.empty((total,), dtype=torch.int32, device=dev)
    for r in range(world):
        if not counts[r]:
            continue
        gi = g_idx[r][:counts[r]] if g_idx is not None else torch.as_tensor(index_of_rank(r, counts[r]), dtype=torch.int64).to(dev)
        t_act[gi] = g_act[r][:counts[r]]
        t_pred[gi] = g_pred[r][:counts[r]]
    if not as_numpy:
        return t_act, t_pred
    return t_act.cpu().numpy(), t_pred.cpu().numpy()


class OrderedRowGather:
    """The product path's only exchange: result rows travel to the writer rank, which writes them in
    work-unit order. Work unit k (a byte range of whole reads, or a batch of fast5 files) belongs to rank k % world;
    round r = units r*world .. r*world + world - 1. Every rank hands its finished unit of round r to put(); a
    COMMUNICATION THREAD per rank runs the collectives (lengths, then the padded byte payload, gathered to rank 0
    over RCCL/xGMI with backend "nccl", or gloo on CPU), so the rank's main thread never blocks on a collective: it
    parses and drives its GPU for round r+1 while round r is on the wire and being written. `depth` bounds the
    finished-but-unsent rounds a rank may hold (back-pressure).

    The rows are gathered as formatted text (the 12 B/site of results plus the six verbatim sampleinfo columns and
    the k-mer that only the owning rank has parsed): rank 0 never touches the other ranks' input bytes.
    Only the communication thread calls into torch.distributed, so the collective order is the same on every rank."""

    def __init__(self, dist, rank: int, world: int, result_file: str, nrounds: int, device=None, depth: int = 4):
        import queue
        import threading
        self.dist, self.rank, self.world, self.nrounds, self.device = dist, rank, world, nrounds, device
        self._q = queue.Queue(maxsize=max(1, depth))
        self._exc = None
        self._put = 0
        self.total_sites = 0
        self.total_errors = 0
        self.failed_ranks = 0
        self.bytes_gathered = 0
        self._wf = open(result_file, "wb") if rank == 0 else None
        self._th = threading.Thread(target=self._run, name="ds-row-gather", daemon=True)
        self._th.start()

    def put(self, data: bytes) -> None:
        """This rank's rows of its next round (b"" when it owns no unit in that round)."""
        if self._exc is not None:
            raise self._exc
        self._put += 1
        self._q.put(bytes(data))

    def close(self, nsites: int = 0, nerrors: int = 0, failed: bool = False):
        """All rounds handed over: wait for the exchange to finish; returns the job-wide (sites, errors) on every rank.
        failed=True (this rank hit an error): its remaining rounds are sent empty so no rank hangs in a collective,
        and EVERY rank's close() raises."""
        while self._put < self.nrounds:
            self.put(b"")
        self._q.put((int(nsites), int(nerrors), 1 if failed else 0))
        self._th.join()
        if self._wf is not None:
            self._wf.close()
        if self._exc is not None:
            raise self._exc
        if self.failed_ranks:
            raise RuntimeError("call_mods failed on %d rank(s); %s is incomplete" % (self.failed_ranks, "the result file"))
        return self.total_sites, self.total_errors

    def _run(self):
        import torch
        try:
            dist, world, rank = self.dist, self.world, self.rank
            dev = self.device if self.device is not None else torch.device("cpu")
            on_gpu = dev.type == "cuda"
            if on_gpu:
                torch.cuda.set_device(dev)
            # staging reused across rounds (grown when a round is bigger): on a GPU the rows go host -> PINNED buffer ->
            # device buffer -> RCCL, and on rank 0 back through one pinned buffer; one host sync per round for the lengths
            cap = 0
            pin = pin_np = dbuf = gbuf = pin_out = None
            lens_dev = torch.zeros(world, dtype=torch.int64, device=dev)
            for _ in range(self.nrounds):
                data = self._q.get()
                n_local = torch.tensor([len(data)], dtype=torch.int64, device=dev)
                dist.all_gather_into_tensor(lens_dev, n_local)
                lens = [int(x) for x in lens_dev.cpu().tolist()]
                nmax = max(lens)
                if nmax == 0:
                    continue
                if nmax > cap:
                    cap = max(nmax, 2 * cap, 1 << 20)
                    pin = torch.empty(cap, dtype=torch.uint8, pin_memory=on_gpu)
                    pin_np = pin.numpy()
                    dbuf = torch.empty(cap, dtype=torch.uint8, device=dev) if on_gpu else pin
                    if rank == 0:
                        gbuf = torch.empty(world * cap, dtype=torch.uint8, device=dev)
                        pin_out = torch.empty(world * cap, dtype=torch.uint8, pin_memory=on_gpu)
                if data:
                    # one copy, bytes -> pinned buffer, through numpy views (torch.frombuffer on immutable bytes warns, and
                    # silencing it with warnings.catch_warnings() in this thread would swap the process-global filter list
                    # under the main thread's feet)
                    pin_np[:len(data)] = np.frombuffer(data, dtype=np.uint8)
                if on_gpu:
                    dbuf[:nmax].copy_(pin[:nmax], non_blocking=True)
                gl = [gbuf[r * nmax:(r + 1) * nmax] for r in range(world)] if rank == 0 else None
                dist.gather(dbuf[:nmax], gl, dst=0)
                if rank == 0:
                    host = pin_out[:world * nmax]
                    host.copy_(gbuf[:world * nmax])                                   # one D2H copy (synchronous into pinned memory)
                    hv = host.numpy()
                    for r in range(world):
                        if lens[r]:
                            self._wf.write(hv[r * nmax:r * nmax + lens[r]].tobytes())
                            self.bytes_gathered += lens[r] if r else 0
                    self._wf.flush()
            counts = self._q.get()
            t = torch.tensor(list(counts), dtype=torch.int64, device=dev)
            dist.all_reduce(t)
            self.total_sites, self.total_errors, self.failed_ranks = int(t[0].item()), int(t[1].item()), int(t[2].item())
        except BaseException as exc:      # surfaced by put() / close() on the main thread
            self._exc = exc
            try:
                while True:
                    self._q.get_nowait()
            except Exception:
                pass
