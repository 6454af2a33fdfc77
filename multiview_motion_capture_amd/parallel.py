"""Frame sharding across the GPUs of one node (one process per GPU, torch.distributed).

Spatial association, DLT and cold-started IK are independent per frame, so frames
shard as contiguous ranges with no data-path collective; one all-gather (RCCL over xGMI with the
"nccl" backend, gloo on CPU) brings every shard's per-frame results to every rank, after which
identities are stitched across shard boundaries on the host (the reference's tracker is a single
sequential pass, motion_capture.py:1062-1116, so this step has no counterpart there).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_frames: int, rank: int, world: int):
    """Contiguous frame range [lo, hi) of ``rank``; ranges differ by at most one frame."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_results(out: Dict[str, torch.Tensor], world: int) -> Dict[str, torch.Tensor]:
    """All-gather every per-frame tensor of ``out`` along dim 0 (equal shard sizes).  One fused
    collective: the tensors are packed into a single byte buffer so RCCL sees one large message."""
    keys = sorted(k for k, v in out.items() if isinstance(v, torch.Tensor))
    if world == 1:
        return out
    flat = [out[k].contiguous().view(torch.uint8).reshape(-1) for k in keys]
    sizes = [f.numel() for f in flat]
    send = torch.cat(flat)
    recv = torch.empty(world * send.numel(), dtype=torch.uint8, device=send.device)
    dist.all_gather_into_tensor(recv, send)
    recv = recv.view(world, -1)
    res, off = {}, 0
    for k, n in zip(keys, sizes):
        t = out[k]
        part = recv[:, off:off + n].contiguous().view(t.dtype).reshape((world * t.shape[0],) + tuple(t.shape[1:]))
        res[k] = part
        off += n
    return res


def stitch_identities(joints_prev: np.ndarray, joints_next: np.ndarray, max_dist=0.5):
    """Match the people of the last frame of one shard to the first frame of the next shard.
    joints_* (K,18,3) with NaN rows for empty slots -> list of (i_prev, i_next) pairs
    (Hungarian assignment on the mean joint distance, pairs farther than max_dist dropped)."""
    from scipy.optimize import linear_sum_assignment
    ok_p = np.nonzero(~np.isnan(joints_prev).any(axis=(1, 2)))[0]
    ok_n = np.nonzero(~np.isnan(joints_next).any(axis=(1, 2)))[0]
    if len(ok_p) == 0 or len(ok_n) == 0:
        return []
    cost = np.linalg.norm(joints_prev[ok_p][:, None] - joints_next[ok_n][None], axis=-1).mean(axis=-1)
    r, c = linear_sum_assignment(cost)
    return [(int(ok_p[i]), int(ok_n[j])) for i, j in zip(r, c) if cost[i, j] <= max_dist]
