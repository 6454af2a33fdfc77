"""Temporal hot path: MvTracker.update_4d (motion_capture.py:873-963) for a batch of independent
chains (sub-sequences), every stage on the GPU and no host synchronisation inside a step.

Per step and chain: live tracklets + the frame's 2-D poses -> match_spatial_time graph -> ALS ->
tracklet-anchored clusters (warm IK from the previous parameters) and 2-D-only clusters (new tracklets,
cold IK); chains without live tracklets take the match_spatial path, exactly as
associate_tracking does (motion_capture.py:829-835).
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch

from . import device as dev
from .pipeline import HotPath

T_WIDE = 16   # tracklet slots of the repair tier (include/mvmc.h: the stitch and the widest association variant hold 16)


def default_caps(n_views: int, p_max: int):
    """(k_max, v_max) that a frame's own size rules out ever exceeding: a new tracklet needs two poses, so at most C P / 2 appear in a
    frame; clusters are disjoint sets of the frame's poses, so one holds at most C P (the reference has neither cap:
    motion_capture.py:417-446, :618-626, :763-808).  The kernels index members by lane: 64 at most."""
    n = n_views * p_max
    return max(1, n // 2), min(n, 64)


class ChainTracker:
    def __init__(self, hp: HotPath, n_chains: int, p_max: int, t_max: int = 8, k_max: Optional[int] = None,
                 v_max: Optional[int] = None, nfev_cold=50, nfev_warm=5):
        d = hp.device
        self.hp, self.B, self.P, self.T = hp, n_chains, p_max, t_max
        C = hp.K.shape[0]
        self.C = C
        k_def, v_def = default_caps(C, p_max)
        self.K = k_max or k_def
        self.V = v_max or v_def
        self.nfev_cold, self.nfev_warm = nfev_cold, nfev_warm
        if 2 * p_max > 16 or t_max > T_WIDE or C * p_max + t_max > 80 or t_max + self.K > 64 or self.V > 64:
            raise ValueError(f"ChainTracker: p_max={p_max}, t_max={t_max}, views={C}: the association kernels hold rank 2 p_max <= 16, "
                             f"t_max <= {T_WIDE} and views x p_max + t_max <= 80 graph nodes (more live tracklets than t_max in a frame is "
                             "reported by check())")
        self.F2 = dev.fmats_from_projections(hp.P)
        B, T = n_chains, t_max
        # The tracker state is ONE device allocation with typed views into it: snapshot() / restore() are one copy each (update_4d
        # saves the state in front of every frame), and the per-frame driver reads it back in one transfer (read_back()).
        # overflow -- per chain: bit 0 cluster / view capacity, bit 1 tracklet table, bit 2 a graph the association kernel could not
        # hold (iters < 0); accumulated on the device, read by check(): a non-zero word voids the chain's results
        layout = (("params", (B, T, 68), torch.float64), ("joints", (B, T, 18, 3), torch.float64), ("meta", (B, T, 4), torch.int32),
                  ("n_tracks", (B,), torch.int32), ("next_id", (B,), torch.int32), ("n_dead", (B,), torch.int32),
                  ("slot_src", (B, T), torch.int32), ("overflow", (B,), torch.int32),
                  # the chain kernel's flag words of step_fused (mvmc_chain_run, n_parts = 1): in the same allocation, so that
                  # read_back() brings state and verdict to the host in ONE transfer
                  ("cflags", (2 * B + 8,), torch.int32))
        offs, total = {}, 0
        for name, shape, dt in layout:
            nbytes = int(torch.Size(shape).numel()) * (8 if dt == torch.float64 else 4)
            offs[name] = (total, nbytes, shape, dt)
            total += (nbytes + 15) & ~15
        self._flat = torch.zeros((total,), dtype=torch.uint8, device=d)
        self._layout = offs
        for name, (o, nb, shape, dt) in offs.items():
            setattr(self, name, self._flat[o:o + nb].view(dt).view(shape))
        self.slot_src.fill_(-1)
        self.frame_idx = torch.arange(B, dtype=torch.int32, device=d)
        self._host = None   # pinned host mirror of _flat (+ the chain kernel's time-out words), allocated by read_back()
        self._fused = None  # workspaces of step_fused (allocated on first use)
        self._fused_args = None   # (key, MvmcChainBuffers) of the last step_fused call: the struct is rebuilt only when a pointer changes
        self._in = None     # pinned host staging + device buffers of one frame's inputs (frame_inputs())
        self._void_pending = False   # step_fused(fold_void=False) left its void words for read_back()
        self.events = None  # set to a list to collect (start, end) CUDA events around every IK launch
        self.als_events = None  # same for the association (ALS) launches of the spatio-temporal graph
        self.assoc_done = None

    def step(self, kps17: torch.Tensor, counts: torch.Tensor, want_debug=False):
        """kps17 (B,C,P,17,3) f64 + counts (B,C) i32 of the current frame of every chain."""
        B, C, P, T, K, V = self.B, self.C, self.P, self.T, self.K, self.V
        hp = self.hp
        has = (self.n_tracks > 0)
        # chains without tracklets: match_spatial (f32 affinity); the others get zero people there
        cnt_sp = torch.where(has[:, None], torch.zeros_like(counts), counts)
        _, S = dev.affinity(kps17, cnt_sp, hp.F, want_D=False)
        sp = dev.als_associate(S, cnt_sp, g_max=P)
        # chains with tracklets: match_spatial_time graph
        W, D, gc = dev.st_affinity(kps17, counts, self.frame_idx, self.joints, self.n_tracks, hp.P, self.F2,
                                   want_D=want_debug)
        if self.als_events is not None:
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
        st = dev.als_associate(W, gc, g_max=max(P, T), want_mats=want_debug)
        if self.als_events is not None:
            a1.record()
            self.als_events.append((a0, a1))
        self.overflow |= (((sp["iters"] < 0) | (st["iters"] < 0)).to(torch.int32) * 4)
        mem, cold, init, status, n_new = dev.track_assign(sp["labels"], sp["n_clusters"], st["labels"],
                                                          st["n_clusters"], counts, self.frame_idx, self.n_tracks,
                                                          self.params, P, K, V, overflow=self.overflow)
        NP = T + K
        if self.assoc_done is not None:   # one-shot marker for run_chains' stream stagger
            self.assoc_done.record()
            self.assoc_done = None
        if self.events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        p, j, info = dev.ik_solve(kps17, hp.P, mem.reshape(B * NP, V), init.reshape(B * NP, 68),
                                  cold.reshape(B * NP), self.nfev_cold, self.nfev_warm, hp.skeleton)
        if self.events is not None:
            e1.record()
            self.events.append((e0, e1))
        p, j = p.reshape(B, NP, 68), j.reshape(B, NP, 18, 3)
        dev.track_commit(status, n_new, p, j, self.params, self.joints, self.meta, self.n_tracks, self.next_id,
                         self.n_dead, K, slot_src=self.slot_src, overflow=self.overflow)
        out = dict(members=mem, status=status, n_new=n_new, ik_params=p, ik_joints=j, ik_info=info.reshape(B, NP, 8))
        if want_debug:
            out.update(D=D, W=W, st=st, sp=sp, group_counts=gc)
        return out


    def frame_inputs(self):
        """Staging for the per-frame driver (MvTracker.update_4d): ONE pinned host buffer and ONE device buffer that hold a frame's
        keypoints (B,C,P,17,3) f64 and counts (B,C) i32 side by side -- the caller fills the NumPy views `kps_np` / `cnt_np`, calls
        upload_inputs() (one asynchronous copy instead of two pageable ones), and hands `kps_d` / `cnt_d` to step_fused() / step().
        The buffers are reused by the next frame: the driver synchronises at the end of every frame (read_back())."""
        if self._in is None:
            B, C, P = self.B, self.C, self.P
            nk, nc = B * C * P * 17 * 3 * 8, B * C * 4
            total = nk + ((nc + 15) & ~15)
            host = torch.empty((total,), dtype=torch.uint8).pin_memory()
            devb = torch.empty((total,), dtype=torch.uint8, device=self._flat.device)
            self._in = dict(host=host, dev=devb,
                            kps_np=host[:nk].view(torch.float64).view(B, C, P, 17, 3).numpy(),
                            cnt_np=host[nk:nk + nc].view(torch.int32).view(B, C).numpy(),
                            kps_d=devb[:nk].view(torch.float64).view(B, C, P, 17, 3),
                            cnt_d=devb[nk:nk + nc].view(torch.int32).view(B, C))
        return self._in

    def upload_inputs(self) -> None:
        self._in["dev"].copy_(self._in["host"], non_blocking=True)

    def step_fused(self, kps17: torch.Tensor, counts: torch.Tensor, fold_void: bool = True):
        """The same frame update as step() in ONE launch (mvmc_chain_run with chain_len 1 on this tracker's state): what
        the per-frame call surface (MvTracker.update_4d) uses.  Sizes outside the chain kernel's arena, or a frame whose graph
        is too large for it, are the caller's to route to step() (ChainTracker.fused_ok, check_chain_flags).
        fold_void=False: the launch's per-chain void words are NOT folded into the tracker's own (one small kernel less per frame) --
        for a caller that ends the frame with read_back(), which then reads them where the launch left them."""
        import ctypes as C
        from . import _cabi
        B, Cn, P, T, K, V = self.B, self.C, self.P, self.T, self.K, self.V
        d = kps17.device
        N, NS, NP = Cn * P, T + Cn * P, T + K
        if self._fused is None:
            f64, i32 = torch.float64, torch.int32
            e = lambda shape, dt: torch.empty(shape, dtype=dt, device=d)
            z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=d)
            self._fused = dict(
                seed_table=dev.als_seed_table(_cabi.MAX_NODES * _cabi.MAX_NODES, d),
                S_sp=e((B, N, N), torch.float32), W_st=e((B, NS, NS), f64), group_counts=e((B, Cn + 1), i32),
                labels_sp=e((B, N), i32), labels_st=e((B, NS), i32), n_clusters_sp=z((B,), i32), n_clusters_st=z((B,), i32),
                iters_sp=z((B,), i32), iters_st=z((B,), i32), members=e((B, NP, V), i32), n_members=z((B, NP), i32),
                cold=e((B, NP), torch.uint8),
                init=e((B, NP, 68), f64), status=e((B, T), i32), n_new=e((B,), i32), ik_params=e((B, NP, 68), f64),
                ik_joints=e((B, NP, 18, 3), f64), ik_info=e((B, NP, 8), f64), ik_scratch=_chain_scratch(B, d),
                out_params=e((B, T, 68), f64), out_joints=e((B, T, 18, 3), f64), out_meta=e((B, T, 4), i32),
                out_n_tracks=e((B,), i32), flags=self.cflags)
        w = self._fused
        # the argument struct: every pointer in it but the frame's inputs belongs to this tracker, and the per-frame driver hands in
        # the same input buffers every frame (frame_inputs()) -- built once, rebuilt when an input pointer changes
        key = (kps17.data_ptr(), counts.data_ptr(), self.nfev_cold, self.nfev_warm)
        if self._fused_args is None or self._fused_args[0] != key:
            t = dict(w, kps17=kps17, counts=counts, Pmats=self.hp.P, Fmats=self.hp.F, F2=self.F2, params=self.params,
                     joints=self.joints, meta=self.meta, n_tracks=self.n_tracks, next_id=self.next_id, n_dead=self.n_dead,
                     slot_src=self.slot_src, out_info=None, out_als_iters=None, out_phase_cycles=None)
            buf = _cabi.MvmcChainBuffers()
            for name, val in dict(n_chains=B, chain_len=1, n_views=Cn, p_max=P, t_max=T, k_max=K, v_max=V,
                                  max_nfev_cold=self.nfev_cold, max_nfev_warm=self.nfev_warm, n_inits=3,
                                  seed_len=w["seed_table"].numel(), n_parts=1, force_big=0, hand_over=0).items():
                setattr(buf, name, int(val))
            for name in _cabi.MvmcChainBuffers._PTRS:
                ten = t[name]
                setattr(buf, name, None if ten is None else ten.data_ptr())
            self._fused_args = (key, buf)
        buf = self._fused_args[1]
        _cabi.check(_cabi.load().mvmc_chain_run(C.byref(self.hp.skeleton), C.byref(buf),
                                                C.c_void_p(torch.cuda.current_stream(d).cuda_stream)), "mvmc_chain_run")
        # the launch zeroes its flag words: fold this frame's per-chain void words into the tracker's own (read by check())
        self._void_pending = not fold_void
        if fold_void:
            self.overflow |= w["flags"][B + 4:2 * B + 4]
        return dict(members=w["members"], n_members=w["n_members"], status=w["status"], n_new=w["n_new"], ik_params=w["ik_params"],
                    ik_joints=w["ik_joints"], ik_info=w["ik_info"], flags=w["flags"], void=w["flags"][B + 4:2 * B + 4], n_chains=B,
                    chain_len=1)

    def check(self) -> None:
        """Raise if a capacity was exceeded since the last call (synchronises), and clear the words: the report is per call, so a
        caller that restores the state it saved before the frame (snapshot / restore) can go on -- MvTracker.update_4d does, with a
        wider table.  The reference has no such caps, so a frame that hits one is not tracked the way the reference would."""
        ov = int(self.overflow.max()) if self.overflow.numel() else 0
        if self._fused is not None:
            B = self.B
            fl = self._fused["flags"][B:B + 4].cpu().tolist()
            self._fused["flags"][B:B + 4].zero_()
            if fl[0]:
                raise RuntimeError("mvmc_chain_run: a hand-over between the workgroups of a chain timed out; results are void")
        self.overflow.zero_()
        if ov:
            what = [m for bit, m in ((1, "a cluster, a member or a view block did not fit (k_max / v_max / the frame's poses)"),
                                     (2, "more than t_max live tracklets"),
                                     (4, "a graph larger than the association kernel holds"),
                                     (8, "internal: a meeting of two IK waves timed out (mvmc_ik_pair.h)")) if ov & bit]
            raise ValueError("ChainTracker: capacity exceeded (" + "; ".join(what) + "): the frame's results are void")

    _STATE = ("params", "joints", "meta", "n_tracks", "next_id", "n_dead", "slot_src")

    def snapshot(self):
        """The tracker state (one device copy): restore() brings it back, e.g. to redo a frame that exceeded a capacity."""
        return self._flat.clone()

    def restore(self, snap) -> None:
        self._flat.copy_(snap)
        self.overflow.zero_()

    def read_back(self):
        """The state on the host after ONE transfer and ONE synchronisation (the per-frame driver's end of frame: check() and four
        tensor reads took six round trips): dict of NumPy views (params, joints, meta, n_tracks, ..., overflow, cflags) of one of two
        pinned buffers (the call after next overwrites it).  Raises like check(); the words that made it raise are cleared on the device
        (nothing is cleared on a frame that went through: the next launch zeroes its own words)."""
        n, B = self._flat.numel(), self.B
        if self._host is None:
            # two pinned mirrors, written alternately: the one NOT written by this call holds the state after the last frame that went
            # through, i.e. the state in front of this one -- what restore_previous() brings back without a per-frame device snapshot
            self._host = [torch.empty((n,), dtype=torch.uint8).pin_memory() for _ in range(2)]
            self._host_good = -1         # index of the mirror that holds the last good state (-1: none yet)
        cur = 1 - self._host_good if self._host_good >= 0 else 0
        h = self._host[cur]
        h[:n].copy_(self._flat, non_blocking=True)          # (state AND the chain kernel's flag words: `cflags` is part of _flat)
        fl = self.cflags[B:2 * B + 4] if self._fused is not None else None
        torch.cuda.current_stream(self._flat.device).synchronize()
        out = {name: h[o:o + nb].view(dt).view(shape).numpy() for name, (o, nb, shape, dt) in self._layout.items()}
        words = out["cflags"][B:2 * B + 4] if fl is not None else None
        ov = int(out["overflow"].max()) if out["overflow"].size else 0
        if words is not None and self._void_pending:
            ov |= int(np.bitwise_or.reduce(words[4:])) if B else 0       # (step_fused(fold_void=False): read where the launch left them)
            self._void_pending = False
        # clear what was set -- on the device only when something WAS set (the next launch zeroes its own words anyway): the common
        # frame ends with one transfer, one synchronisation and no further kernel
        if ov:
            self.overflow.zero_()
        if words is not None and int(words[0]):
            fl.zero_()
            raise RuntimeError("mvmc_chain_run: a hand-over between the workgroups of a chain timed out; results are void")
        if ov:
            what = [m for bit, m in ((1, "a cluster, a member or a view block did not fit (k_max / v_max / the frame's poses)"),
                                     (2, "more than t_max live tracklets"),
                                     (4, "a graph larger than the association kernel holds"),
                                     (8, "internal: a meeting of two IK waves timed out (mvmc_ik_pair.h)")) if ov & bit]
            raise ValueError("ChainTracker: capacity exceeded (" + "; ".join(what) + "): the frame's results are void")
        self._host_good = cur
        return out

    @property
    def has_previous(self) -> bool:
        """Whether read_back() has left a host mirror of the state after the last good frame (restore_previous())."""
        return self._host is not None and self._host_good >= 0

    def restore_previous(self) -> None:
        """The state after the last frame that read_back() returned for -- the state in front of a frame that has just failed --
        back onto the device, from the pinned mirror (the per-frame driver then needs no device snapshot in front of every frame)."""
        self._flat.copy_(self._host[self._host_good][:self._flat.numel()], non_blocking=True)
        self.overflow.zero_()

    def widened(self, t_max: int) -> "ChainTracker":
        """A tracker with t_max tracklet slots (> the present number) holding this tracker's state."""
        w = ChainTracker(self.hp, self.B, self.P, t_max, nfev_cold=self.nfev_cold, nfev_warm=self.nfev_warm)
        T = self.T
        w.params[:, :T], w.joints[:, :T], w.meta[:, :T], w.slot_src[:, :T] = self.params, self.joints, self.meta, self.slot_src
        w.n_tracks.copy_(self.n_tracks); w.next_id.copy_(self.next_id); w.n_dead.copy_(self.n_dead)
        w.frame_idx = self.frame_idx
        return w

    def narrowed(self, t_max: int) -> "ChainTracker":
        """The inverse of widened(): a tracker with t_max slots holding this one's first t_max (the caller has checked that no chain
        has more live tracklets than that) -- back on the tables the chain kernel runs on once a crowded scene has thinned out."""
        if int(self.n_tracks.max()) > t_max:      # (live tracklets occupy the first n_tracks slots: track_commit compacts the table)
            raise ValueError(f"ChainTracker.narrowed: a chain has more than {t_max} live tracklets")
        n = ChainTracker(self.hp, self.B, self.P, t_max, nfev_cold=self.nfev_cold, nfev_warm=self.nfev_warm)
        n.params.copy_(self.params[:, :t_max]); n.joints.copy_(self.joints[:, :t_max]); n.meta.copy_(self.meta[:, :t_max])
        n.slot_src.copy_(self.slot_src[:, :t_max])
        n.n_tracks.copy_(self.n_tracks); n.next_id.copy_(self.next_id); n.n_dead.copy_(self.n_dead)
        n.frame_idx = self.frame_idx
        return n

    @property
    def fused_ok(self) -> bool:
        """Whether this tracker's padded sizes fit the chain kernel (include/mvmc.h: mvmc_chain_run)."""
        N = self.C * self.P
        small = N <= 40 and self.T + N <= 48 and self.T <= 8
        big = N <= 64 and self.T + N <= 80 and self.T <= T_WIDE
        return (small or big) and self.P <= 8 and self.C <= 16 and self.T + self.K <= 64


def run_chains(hp: HotPath, kps: torch.Tensor, counts: Optional[torch.Tensor], chain_len: int, t_max=8,
               nfev_cold=50, nfev_warm=5, events=None, want_info=False, n_groups=1, als_events=None, k_max: Optional[int] = None,
               v_max: Optional[int] = None):
    """Whole shard: frames [c*L, (c+1)*L) form chain c (F must be a multiple of L).  Returns per-frame
    tracklet tables: params (F,T,68), joints (F,T,18,3), meta (F,T,4), n_tracks (F).

    n_groups > 1 splits the chains into that many groups, each advanced on its own HIP stream: inside a chain the
    stages of a frame are strictly sequential (association -> assignment -> IK -> commit -> next frame), and the
    association solver is a long dependent chain on few waves, so one group's association runs in the shadow of
    another group's IK launch.  Results do not depend on n_groups (chains are independent)."""
    F, C, P = kps.shape[:3]
    L = chain_len
    if F % L:
        raise ValueError("run_chains: the frame count must be a multiple of the chain length")
    B = F // L
    G = max(1, min(int(n_groups), B))
    kps17, cnt = dev.ingest(kps, counts)
    k4 = kps17.view(B, L, C, P, 17, 3)
    c4 = cnt.view(B, L, C)
    d = kps.device
    out_p = torch.empty((B, L, t_max, 68), dtype=torch.float64, device=d)
    out_j = torch.empty((B, L, t_max, 18, 3), dtype=torch.float64, device=d)
    out_m = torch.empty((B, L, t_max, 4), dtype=torch.int32, device=d)
    out_n = torch.empty((B, L), dtype=torch.int32, device=d)
    n_dead = torch.empty((B,), dtype=torch.int32, device=d)
    next_id = torch.empty((B,), dtype=torch.int32, device=d)
    overflow = torch.empty((B,), dtype=torch.int32, device=d)
    bounds = [B * g // G for g in range(G + 1)]
    main = torch.cuda.current_stream(d)
    streams = [main] if G == 1 else [torch.cuda.Stream(device=d) for _ in range(G)]
    ready = torch.cuda.Event()
    ready.record(main)
    trackers, infos = [], [[] for _ in range(G)]
    for g in range(G):
        with torch.cuda.stream(streams[g]):
            streams[g].wait_event(ready)
            tr = ChainTracker(hp, bounds[g + 1] - bounds[g], P, t_max, k_max=k_max, v_max=v_max, nfev_cold=nfev_cold, nfev_warm=nfev_warm)
            tr.events = events
            tr.als_events = als_events
            trackers.append(tr)
    # stagger: group g starts once group g-1 has finished the association of its first frame, so that from then on
    # the association launches of one group and the IK launches of another alternate instead of colliding
    stagger = [torch.cuda.Event() for _ in range(G - 1)]
    for g in range(G - 1):
        trackers[g].assoc_done = stagger[g]
    for t in range(L):
        for g in range(G):
            b0, b1 = bounds[g], bounds[g + 1]
            with torch.cuda.stream(streams[g]):
                tr = trackers[g]
                if t == 0 and g > 0:
                    streams[g].wait_event(stagger[g - 1])
                o = tr.step(k4[b0:b1, t].contiguous(), c4[b0:b1, t].contiguous())
                if want_info:
                    infos[g].append(o["ik_info"])
                out_p[b0:b1, t], out_j[b0:b1, t], out_m[b0:b1, t], out_n[b0:b1, t] = tr.params, tr.joints, tr.meta, tr.n_tracks
    for g in range(G):
        with torch.cuda.stream(streams[g]):
            n_dead[bounds[g]:bounds[g + 1]] = trackers[g].n_dead
            next_id[bounds[g]:bounds[g + 1]] = trackers[g].next_id
            overflow[bounds[g]:bounds[g + 1]] = trackers[g].overflow
        if streams[g] is not main:
            main.wait_stream(streams[g])
    res = dict(params=out_p.view(F, t_max, 68), joints=out_j.view(F, t_max, 18, 3), meta=out_m.view(F, t_max, 4),
               n_tracks=out_n.view(F), n_dead=n_dead, next_id=next_id, overflow=overflow)
    if want_info:
        res["ik_info"] = torch.cat([torch.stack(i, 1) for i in infos], 0)
    return res


def run_chains_fused(hp: HotPath, kps: torch.Tensor, counts: Optional[torch.Tensor], chain_len: int, t_max: Optional[int] = None,
                     nfev_cold=50, nfev_warm=5, want_info=False, k_max: Optional[int] = None, v_max: Optional[int] = None,
                     parts: Optional[int] = None, kernel_events: Optional[list] = None, force_big: bool = False,
                     hand_over: Optional[str] = None):
    """run_chains in ONE launch (mvmc_chain_run): a persistent workgroup per chain runs graph -> ALS -> assignment ->
    IK -> commit for the chain's frames, so every chain advances at its own pace instead of waiting, stage by stage,
    for the slowest member of every launch.  Same device code and the same results as run_chains.
    parts > 1 (a divisor of chain_len): every chain is run by that many workgroups, one frame range after the other
    (hand-over through device flags), which lets the hardware dispatcher even out the load when the number of chains is
    not a multiple of the number of workgroup slots.  check_chain_flags(res) tells whether the run is valid.
    kernel_events: a list that receives the (start, end) torch.cuda.Event pair recorded right around the kernel launch.
    hand_over: "ticket" (default: a workgroup draws a ticket when it starts, ticket = part * n_chains + chain; a part's predecessor
    holds a lower ticket, so it has started: no assumption about the order of dispatch), "static" (the same mapping by block index:
    relies on in-order dispatch, bounded wait) or "queue" (ready queue: a freed slot goes to the chain that has been ready longest;
    no assumption either, ~3 % slower).  Same results bit for bit."""
    import ctypes as C
    from . import _cabi
    F, Cn, P = kps.shape[:3]
    L = chain_len
    if F % L:
        raise ValueError("run_chains_fused: the frame count must be a multiple of the chain length")
    B = F // L
    if parts is None:
        parts = L   # one workgroup per chain-frame: the finest hand-over, the best balance
    # tracklet slots: 8 on the SMALL layout (views x people <= 40: its association variants hold rank 16), 16 on the BIG one (C8 P8),
    # whose workgroup takes a frame with a ninth tracklet (rank 18, 73 nodes) through its generic association variant in place
    T = t_max if t_max is not None else (T_WIDE if Cn * P > 40 else 8)
    k_def, v_def = default_caps(Cn, P)
    K = k_max or k_def
    V = v_max or v_def
    N, NS, NP = Cn * P, T + Cn * P, T + K
    kps17, cnt = dev.ingest(kps, counts)
    d = kps.device
    F2 = dev.fmats_from_projections(hp.P)
    seed = dev.als_seed_table(_cabi.MAX_NODES * _cabi.MAX_NODES, d)
    f64, i32 = torch.float64, torch.int32
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=d)
    e = lambda shape, dt: torch.empty(shape, dtype=dt, device=d)
    t = dict(
        kps17=kps17, counts=cnt, Pmats=hp.P, Fmats=hp.F, F2=F2, seed_table=seed,
        params=z((B, T, 68), f64), joints=z((B, T, 18, 3), f64), meta=z((B, T, 4), i32), n_tracks=z((B,), i32),
        next_id=z((B,), i32), n_dead=z((B,), i32), slot_src=torch.full((B, T), -1, dtype=i32, device=d),
        S_sp=e((B, N, N), torch.float32), W_st=e((B, NS, NS), f64), group_counts=e((B, Cn + 1), i32),
        labels_sp=e((B, N), i32), labels_st=e((B, NS), i32), n_clusters_sp=z((B,), i32), n_clusters_st=z((B,), i32),
        iters_sp=z((B,), i32), iters_st=z((B,), i32), members=e((B, NP, V), i32), n_members=z((B, NP), i32),
        cold=e((B, NP), torch.uint8),
        init=e((B, NP, 68), f64), status=e((B, T), i32), n_new=e((B,), i32), ik_params=e((B, NP, 68), f64),
        ik_joints=e((B, NP, 18, 3), f64), ik_info=e((B, NP, 8), f64), ik_scratch=_chain_scratch(B, d),
        out_params=e((F, T, 68), f64), out_joints=e((F, T, 18, 3), f64), out_meta=e((F, T, 4), i32), out_n_tracks=e((F,), i32),
        out_info=e((F, NP, 8), f64) if want_info else None, out_als_iters=e((F,), i32) if want_info else None,
        flags=z((B * (parts + 1) + 8,), torch.int32),
        out_phase_cycles=e((B, 8), f64) if want_info else None)
    if parts > 1 and L % parts:
        raise ValueError("run_chains_fused: parts must divide the chain length")
    if hand_over is None:
        hand_over = "ticket"
    if hand_over not in ("static", "queue", "ticket"):
        raise ValueError("run_chains_fused: hand_over must be 'ticket', 'static' or 'queue'")
    buf = _cabi.MvmcChainBuffers()
    for name, val in dict(n_chains=B, chain_len=L, n_views=Cn, p_max=P, t_max=T, k_max=K, v_max=V, max_nfev_cold=nfev_cold,
                          max_nfev_warm=nfev_warm, n_inits=3, seed_len=seed.numel(), n_parts=parts, force_big=int(force_big),
                          hand_over={"static": 0, "queue": 1, "ticket": 2}[hand_over]).items():
        setattr(buf, name, int(val))
    for name, ten in t.items():
        setattr(buf, name, None if ten is None else ten.data_ptr())
    if kernel_events is not None:
        k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k0.record()
    _cabi.check(_cabi.load().mvmc_chain_run(C.byref(hp.skeleton), C.byref(buf),
                                            C.c_void_p(torch.cuda.current_stream(d).cuda_stream)), "mvmc_chain_run")
    if kernel_events is not None:
        k1.record()
        kernel_events.append((k0, k1))
    # void_words: {a hand-over timed out, a graph too large for the layout's association, a capacity exceeded} -- non-zero = the step is
    # void unless repair_chains clears it; parallel.run_sharded sends them along, so every rank learns of a void step from the gathered
    # messages instead of each rank reading its own words back before the collective
    res = dict(params=t["out_params"], joints=t["out_joints"], meta=t["out_meta"], n_tracks=t["out_n_tracks"],
               n_dead=t["n_dead"], next_id=t["next_id"], flags=t["flags"], void=t["flags"][B + 4:2 * B + 4],
               void_words=t["flags"][B:B + 3], n_chains=B, chain_len=L, _keepalive=t)
    if want_info:
        res["ik_info"] = t["out_info"].view(B, L, NP, 8)
        res["als_iters"] = t["out_als_iters"].view(B, L)
        res["phase_cycles"] = t["out_phase_cycles"]
    return res


def check_chain_flags(res) -> None:
    """Raise if a run_chains_fused / run_chains result is void (synchronises): a hand-over timed out, a frame's graph was larger
    than the chain kernel's association variant holds, or a capacity (t_max; k_max / v_max when the caller passed smaller ones than
    default_caps) was exceeded -- the reference has no such caps.  repair_chains re-runs the chains concerned with wider tables."""
    if "flags" not in res:      # run_chains: per-chain words {1: clusters / views, 2: tracklet table, 4: graph too large}
        ov = int(res["overflow"].max()) if res["overflow"].numel() else 0
        if ov:
            raise ValueError(f"run_chains: capacity exceeded (word {ov}: 1 = k_max / v_max, 2 = t_max, 4 = graph larger than the "
                             "association kernel holds): results are void")
        return
    B = res["n_chains"]
    fl = res["flags"][B:B + 4].cpu().tolist()
    if fl[0]:
        raise RuntimeError("mvmc_chain_run: a hand-over between the workgroups of a chain timed out; results are void")
    void = res["flags"][B + 4:2 * B + 4]
    ov = int(void.max()) if B else 0
    if ov & 8:
        raise RuntimeError("mvmc_chain_run: a meeting of two IK waves timed out (mvmc_ik_pair.h); results are void")
    if ov & 4:
        raise ValueError("mvmc_chain_run: a frame's graph has more nodes than the chain kernel's layout supports (small layout: 24 "
                         "without, 32 with tracklets); repair_chains / run_chains take such data")
    if ov:
        raise ValueError("mvmc_chain_run: capacity exceeded (" + ("a cluster, a member or a view block did not fit; " if ov & 1 else "")
                         + ("more than t_max live tracklets" if ov & 2 else "") + f") in {int((void != 0).sum())} chain(s): their results are void")


def repair_chains(hp: HotPath, kps: torch.Tensor, counts: Optional[torch.Tensor], res, nfev_cold=50, nfev_warm=5,
                  t_wide: int = T_WIDE, big_first: bool = True) -> int:
    """What the reference does where the chain kernel's fixed tables end (it has no caps at all): the chains whose void word is set --
    more live tracklets than t_max, a graph beyond the layout's association variant -- are run again through the per-stage entry
    points with t_wide tracklet slots (association on up to 80 nodes, rank 32), and their rows of ``res`` (run_chains_fused's result,
    same kps / counts) are replaced; the per-frame tables are widened to the slots the repaired chains need.  Synchronises (it reads
    the void words); returns the number of chains repaired.  Raises if a hand-over timed out or a chain exceeds the repair tier too.
    big_first: chains voided by the SMALL layout go through the chain kernel's BIG layout first (one launch), see below."""
    B, L = res["n_chains"], res["chain_len"]
    fl = res["flags"][B:B + 4].cpu().tolist()
    if fl[0]:
        raise RuntimeError("mvmc_chain_run: a hand-over between the workgroups of a chain timed out; results are void")
    if fl[2] & 8:
        raise RuntimeError("mvmc_chain_run: a meeting of two IK waves timed out (mvmc_ik_pair.h); results are void")
    if not (fl[1] or fl[2]):
        return 0
    idx = torch.nonzero(res["void"]).flatten()
    n = int(idx.numel())
    if n == 0:
        return 0
    C, P = kps.shape[1:3]
    k5 = kps.view(B, L, *kps.shape[1:])[idx].reshape(n * L, *kps.shape[1:]).contiguous()
    c5 = None if counts is None else counts.view(B, L, C)[idx].reshape(n * L, C).contiguous()
    sub = None
    if big_first and os.environ.get("MVMC_REPAIR_BIG_FIRST", "1") != "0" and C * P <= 40 and t_wide <= T_WIDE and res["params"].shape[1] <= 8:
        # The chains came from the SMALL layout (views x people <= 40: graphs of <= 32 nodes, 8 tracklet slots, <= 6 views per cluster).
        # What voids there -- a crowded frame of 5 x 6 or 7 x 5, a ninth tracklet -- is inside the BIG layout's tables (80 nodes,
        # 16 slots, 8 views): the same persistent kernel in its 512-thread form takes all of them in ONE launch (bit-identical to the
        # per-stage path, tests/test_gpu_chain_fused.py), which matters when a geometry voids EVERY chain (C5 P6 with everybody in view:
        # measured in tests/test_gpu_capacity_flags.py).  What is beyond that too falls through to the per-stage entry points below.
        big = run_chains_fused(hp, k5, c5, L, t_max=t_wide, nfev_cold=nfev_cold, nfev_warm=nfev_warm, force_big=True)
        bfl = big["flags"][n:n + 4].cpu().tolist()
        if not bfl[0] and int(big["void"].max()) == 0:
            sub = dict(params=big["params"], joints=big["joints"], meta=big["meta"], n_tracks=big["n_tracks"], n_dead=big["n_dead"],
                       next_id=big["next_id"])
    if sub is None:
        sub = run_chains(hp, k5, c5, L, t_max=t_wide, nfev_cold=nfev_cold, nfev_warm=nfev_warm)
        ov = int(sub["overflow"].max())
        if ov:
            raise ValueError(f"repair_chains: a chain exceeds the repair tier as well (word {ov}: 2 = more than {t_wide} live tracklets, "
                             "4 = a graph of more than 80 nodes)")
    T = res["params"].shape[1]
    need = int(sub["n_tracks"].max())
    if need > T:   # widen the per-frame tables (rare: the repaired chains hold more tracklets than the tables have slots)
        F = B * L
        for k, tail in (("params", (68,)), ("joints", (18, 3)), ("meta", (4,))):
            wide = torch.zeros((F, t_wide) + tail, dtype=res[k].dtype, device=res[k].device)
            wide[:, :T] = res[k]
            res[k] = wide
        T = t_wide
    for k in ("params", "joints", "meta"):
        tail = res[k].shape[2:]
        res[k].view(B, L, T, *tail)[idx] = sub[k].view(n, L, t_wide, *tail)[:, :, :T]
    res["n_tracks"].view(B, L)[idx] = sub["n_tracks"].view(n, L)
    res["n_dead"][idx] = sub["n_dead"]
    res["next_id"][idx] = sub["next_id"]
    res["void"][idx] = 0
    res["flags"][B + 1:B + 3] = 0
    res["repaired"] = idx
    return n


_CHAIN_SCRATCH = {}


def _chain_scratch(n_chains: int, d) -> torch.Tensor:
    key = (str(d), torch.cuda.current_stream(d).cuda_stream)
    buf = _CHAIN_SCRATCH.get(key)
    if buf is None or buf.shape[0] < n_chains:
        from . import _cabi
        buf = torch.empty((n_chains, 8, _cabi.IK_SCRATCH_DOUBLES), dtype=torch.float64, device=d)
        _CHAIN_SCRATCH[key] = buf
    return buf
