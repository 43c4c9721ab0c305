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


def gather_results(act, pred, index, dist=None, dst: int = 0, device=None, as_numpy: bool = True):
    """Gather (act float32[n_i,C], pred int32[n_i], index int64[n_i]) from all ranks to `dst`.

    Ragged: every rank pads to the max shard length; returns on dst (act, pred) re-ordered so that
    row j is the site with global index j, and (None, None) elsewhere. `dist` = torch.distributed
    (already initialised) or None for single-process. as_numpy=False keeps the re-ordered results as tensors on
    `device` (no device-to-host copy inside the call: what bench.py times is the RCCL exchange itself)."""
    import torch
    act_t = torch.as_tensor(act, dtype=torch.float32)
    pred_t = torch.as_tensor(pred, dtype=torch.int32)
    idx_t = torch.as_tensor(index, dtype=torch.int64)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        order = torch.argsort(idx_t.cpu())
        return act_t.cpu()[order].numpy(), pred_t.cpu()[order].numpy()
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else act_t.device
    C = act_t.shape[1]
    n_local = torch.tensor([act_t.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    counts = [int(c.item()) for c in counts]
    nmax = max(max(counts), 1)
    # one packed float64 buffer per rank: [nmax, C + 2] = act | pred | index (exact for int < 2^53)
    packed = torch.zeros((nmax, C + 2), dtype=torch.float64, device=dev)
    n = act_t.shape[0]
    if n:
        packed[:n, :C] = act_t.to(dev, torch.float64)
        packed[:n, C] = pred_t.to(dev, torch.float64)
        packed[:n, C + 1] = idx_t.to(dev, torch.float64)
    gl = [torch.empty_like(packed) for _ in range(world)] if rank == dst else None
    dist.gather(packed, gl, dst=dst)
    if rank != dst:
        return None, None
    total = sum(counts)
    if not as_numpy:
        t_act = torch.empty((total, C), dtype=torch.float32, device=dev)
        t_pred = torch.empty((total,), dtype=torch.int32, device=dev)
        for r in range(world):
            blk = gl[r][:counts[r]]
            gi = blk[:, C + 1].to(torch.int64)
            t_act[gi] = blk[:, :C].to(torch.float32)
            t_pred[gi] = blk[:, C].to(torch.int32)
        return t_act, t_pred
    out_act = np.empty((total, C), np.float32)
    out_pred = np.empty((total,), np.int32)
    for r in range(world):
        blk = gl[r][:counts[r]].cpu().numpy()
        gi = blk[:, C + 1].astype(np.int64)
        out_act[gi] = blk[:, :C].astype(np.float32)
        out_pred[gi] = blk[:, C].astype(np.int32)
    return out_act, out_pred
