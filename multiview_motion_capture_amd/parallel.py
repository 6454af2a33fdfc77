"""Frame sharding across the GPUs of one node (one process per GPU, torch.distributed).

A sequence is cut into chains (sub-sequences of ``chain_len`` frames that cold-start); chains are independent,
so contiguous chain ranges go to the ranks with **no data-path collective**.  Each rank packs its results into one message (live
tracklets only, float32: ~0.5 KB per tracklet-frame) and stitches the identities across ITS OWN chain
boundaries while it packs; ONE all-gather (RCCL over xGMI under the "nccl" backend, gloo on CPU) brings every shard's message to every
rank, and a device kernel matches the world - 1 boundaries between shards and turns the shard-local identities into global ones -- the
result of one pass over all chain boundaries, at a per-rank cost that does not grow with the number of ranks.  The reference has no
counterpart: its tracker is a single sequential pass (motion_capture.py:1062-1116).

Pack, gather and stitch are issued on a communication stream behind an event, so they overlap the next step's compute.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, Optional

import numpy as np
import torch
import torch.distributed as dist

ID_CAP = 16          # local identities per chain the stitch has room for
MAX_DIST = 0.5       # metres: pairs farther apart (mean joint distance) are not the same person
T_MSG = 16           # tracklet slots per frame of a message: the widest table any rank can have (tracker.T_WIDE), the SAME on every
                     # rank -- a rank whose repair tier widened its tables must not change the size of what it sends
FORCE_COLLECTIVE = False   # tests: issue the all-gather at world size 1 too (the RCCL path exercised on a one-GPU box)


def shard_range(n_units: int, rank: int, world: int):
    """Contiguous range [lo, hi) of ``rank``; ranges differ by at most one unit (units = chains, or frames)."""
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def chains_cap(n_chains_total: int, world: int) -> int:
    """Chains a message must have room for: the largest shard."""
    return -(-n_chains_total // world)


def message_words(b_cap: int, t_max: int, row_cap: int, id_cap: int = ID_CAP) -> int:
    from . import _cabi
    n = int(_cabi.load().mvmc_pack_message_words(int(b_cap), int(t_max), int(row_cap), int(id_cap)))
    if n < 0:
        raise ValueError("message_words: unsupported sizes")
    return n


def pack_tracks(out: Dict[str, torch.Tensor], next_id: torch.Tensor, chain_len: int, b_cap: int, row_cap: int,
                t_msg: Optional[int] = None, void_words: Optional[torch.Tensor] = None, max_dist: float = MAX_DIST,
                id_cap: int = ID_CAP) -> torch.Tensor:
    """The shard's message (int32 words, include/mvmc.h: mvmc_pack_tracks) from run_chains_fused's result, with the shard's own chain
    boundaries stitched (the message's ``local`` section); asynchronous on the current stream.  ``t_msg``: slots per frame of the
    message (default: the tables'); ``void_words``: device words whose non-zero state travels with the message (the chain kernel's
    time-out / capacity words: tracker.void_words)."""
    from . import _cabi
    from .device import _p, _stream
    F, T = out["params"].shape[:2]
    t_msg = T if t_msg is None else int(t_msg)
    d = out["params"].device
    lib = _cabi.load()
    msg = torch.empty((message_words(b_cap, t_msg, row_cap, id_cap),), dtype=torch.int32, device=d)
    work = torch.empty((int(lib.mvmc_pack_work_words(F, int(chain_len), int(id_cap))),), dtype=torch.int32, device=d)
    if void_words is not None and (void_words.dtype != torch.int32 or not void_words.is_contiguous()):
        raise ValueError("pack_tracks: void_words must be contiguous int32")
    _cabi.check(lib.mvmc_pack_tracks(_p(out["params"]), _p(out["joints"]), _p(out["meta"]), _p(out["n_tracks"]), _p(next_id), F,
                                     int(chain_len), T, t_msg, int(b_cap), int(row_cap), int(id_cap), float(max_dist),
                                     _p(void_words), 0 if void_words is None else int(void_words.numel()), _p(work), _p(msg), _stream()),
                "mvmc_pack_tracks")
    if d.type == "cuda":
        work.record_stream(torch.cuda.current_stream(d))
    return msg


def stitch_chains(messages: torch.Tensor, b_cap: int, t_max: int, row_cap: int, max_dist: float = MAX_DIST, id_cap: int = ID_CAP):
    """messages (world, words) int32 on the device -> dict(gid (Btot_cap, id_cap), match (Btot_cap, T), info (4)); asynchronous."""
    from . import _cabi
    from .device import _p, _stream
    world, words = messages.shape
    d = messages.device
    cap = world * b_cap
    lib = _cabi.load()
    gid = torch.full((cap, id_cap), -1, dtype=torch.int32, device=d)
    match = torch.full((cap, t_max), -1, dtype=torch.int32, device=d)
    info = torch.zeros((4,), dtype=torch.int32, device=d)
    work = torch.empty((int(lib.mvmc_stitch_work_words(world, int(t_max), int(id_cap))),), dtype=torch.int32, device=d)
    _cabi.check(lib.mvmc_stitch_chains(_p(messages), C.c_longlong(words), world, int(b_cap), int(t_max), int(row_cap), int(id_cap),
                                       float(max_dist), cap, _p(gid), _p(match), _p(info), _p(work), _stream()),
                "mvmc_stitch_chains")
    return dict(gid=gid, match=match, info=info, _work=work)


def all_gather_messages(msg: torch.Tensor, world: int) -> torch.Tensor:
    """(world, words) from every rank's (words,) message: one collective."""
    if world == 1 and not (FORCE_COLLECTIVE and dist.is_initialized()):
        return msg.view(1, -1)
    if dist.get_backend() == "nccl":
        out = torch.empty((world, msg.numel()), dtype=msg.dtype, device=msg.device)
        dist.all_gather_into_tensor(out.view(-1), msg)
        return out
    # gloo (CPU tests, or several ranks sharing one GPU): host tensors
    host = msg.cpu()
    parts = [torch.empty_like(host) for _ in range(world)]
    dist.all_gather(parts, host)
    return torch.stack(parts).to(msg.device)


def unpack_message(msg: np.ndarray, b_cap: int, t_max: int, row_cap: int, id_cap: int = ID_CAP) -> dict:
    """Host view of one message (int32 words): header, ids, bounds (B,2,T,56), rows, local stitch -- for consumers and tests."""
    msg = np.ascontiguousarray(msg).view(np.int32)
    h = msg[:8]
    o_ids = 8
    o_b = o_ids + b_cap
    o_r = o_b + b_cap * 2 * t_max * 56
    n_rows = int(h[3])
    bounds = msg[o_b:o_r].reshape(b_cap, 2, t_max, 56)
    rows = msg[o_r:o_r + n_rows * 128].reshape(n_rows, 128)
    o_l = o_r + row_cap * 128
    local = {}
    if msg.size >= o_l + 8 + b_cap * (t_max + id_cap):
        lh = msg[o_l:o_l + 8]
        local = dict(local_roots=int(lh[0]), local_pairs=int(lh[1]), local_error=int(lh[2]), void_word=int(lh[3]),
                     lmatch=msg[o_l + 8:o_l + 8 + b_cap * t_max].reshape(b_cap, t_max)[:int(h[0])].copy(),
                     lgid=msg[o_l + 8 + b_cap * t_max:o_l + 8 + b_cap * (t_max + id_cap)].reshape(b_cap, id_cap)[:int(h[0])].copy())
    return dict(**local, n_chains=int(h[0]), chain_len=int(h[1]), t_max=int(h[2]), n_rows=n_rows, rows_wanted=int(h[4]), n_frames=int(h[6]),
                ids=msg[o_ids:o_ids + int(h[0])].copy(), bound_ids=bounds[:int(h[0]), :, :, 0].copy(),
                bound_joints=bounds[:int(h[0]), :, :, 1:55].copy().view(np.float32).reshape(int(h[0]), 2, t_max, 18, 3),
                row_meta=rows[:, :6].copy(), row_joints=rows[:, 6:60].copy().view(np.float32).reshape(n_rows, 18, 3),
                row_params=rows[:, 60:128].copy().view(np.float32))


def run_sharded(compute: Callable[[], Dict[str, torch.Tensor]], chain_len: int, n_chains_total: int, rank: int, world: int,
                rows_per_frame: int, comm_stream=None, pack: Optional[Callable] = None, stitch: Optional[Callable] = None,
                gather: Optional[Callable] = None, max_dist: float = MAX_DIST, timing: bool = False, t_msg: Optional[int] = None):
    """One step of the sharded path on this rank: compute() (the shard's chains: tracker.run_chains_fused) -> pack -> ONE all-gather ->
    stitch.  ``rows_per_frame`` sizes the message (live tracklets per frame it has room for: the people in the scene; an overflow is
    reported in info[2], never silent).  ``t_msg``: tracklet slots per frame of the message -- every rank must use the same value,
    whatever its own tables look like (a rank whose repair tier widened them included): pass T_MSG when ranks can differ; None = the
    tables' own width (single rank, or tables of one fixed width everywhere).  If compute()'s result carries ``void_words`` (device
    words of the step's validity) they travel in the message and every rank's stitch reports them (info[2] bit 2).
    pack / stitch / gather default to the device kernels and torch.distributed; the CPU tests inject host implementations of the same
    message format.  Returns dict(local=compute's result, messages (world, words),
    gid, match, info, done=event or None, tail_events = [start, packed, gathered, stitched] with timing=True).  With a CUDA
    ``comm_stream`` the pack/gather/stitch tail runs there, behind an event."""
    pack = pack or pack_tracks
    stitch = stitch or stitch_chains
    gather = gather or all_gather_messages
    out = compute()
    F, T = out["params"].shape[:2]
    if t_msg is not None:
        if T > t_msg:
            raise ValueError(f"run_sharded: tables of {T} slots do not fit messages of t_msg = {t_msg}")
        T = int(t_msg)
    b_cap = chains_cap(n_chains_total, world)
    lo, hi = shard_range(n_chains_total, rank, world)
    if F != (hi - lo) * chain_len:
        raise ValueError(f"run_sharded: rank {rank} computed {F} frames, its shard is chains [{lo}, {hi}) x {chain_len}")
    row_cap = b_cap * chain_len * rows_per_frame
    use_stream = comm_stream is not None and out["params"].is_cuda
    if use_stream:
        ready = torch.cuda.Event()
        ready.record()
        ctx = torch.cuda.stream(comm_stream)
        comm_stream.wait_event(ready)
    else:
        import contextlib
        ctx = contextlib.nullcontext()
    with ctx:
        t_tail = None
        if use_stream and timing:
            t_tail = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            t_tail[0].record()
        kw = {}
        if t_msg is not None:
            kw["t_msg"] = T
        if out.get("void_words") is not None:
            kw["void_words"] = out["void_words"]
        msg = pack(out, out["next_id"], chain_len, b_cap, row_cap, **kw)
        if t_tail: t_tail[1].record()
        msgs = gather(msg, world)
        if t_tail: t_tail[2].record()
        st = stitch(msgs, b_cap, T, row_cap, max_dist)
        if t_tail: t_tail[3].record()
        done = None
        if use_stream:
            done = torch.cuda.Event()
            done.record()
            for t in (msg, msgs) + tuple(v for v in st.values() if isinstance(v, torch.Tensor)) + tuple(
                    v for v in out.values() if isinstance(v, torch.Tensor)):
                t.record_stream(comm_stream)
    return dict(local=out, message=msg, messages=msgs, b_cap=b_cap, row_cap=row_cap, done=done, tail_events=t_tail,
                **{k: v for k, v in st.items()})


def check_stitch_info(res) -> None:
    """Raise if the stitched result is void.  Synchronises with the tail first: ``info`` is written on the communication stream, and a
    copy on the current stream is not ordered against it."""
    if res.get("done") is not None:
        res["done"].synchronize()
    info = res["info"].cpu().tolist()
    if info[2] & 4:
        raise RuntimeError("stitch: the compute step of some rank was void (a hand-over time-out or an exceeded capacity of the chain "
                           "kernel, recorded in that rank's message)")
    if info[2] & 1:
        raise RuntimeError("stitch: a message overflowed its row capacity or a chain has more local identities than ID_CAP")
    if info[2]:
        raise RuntimeError("stitch: an assignment did not terminate (error word %d)" % info[2])
