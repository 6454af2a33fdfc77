"""Batched device entry points: torch CUDA tensors in, torch CUDA tensors out.

Thin, allocation-only wrappers over the C ABI (include/mvmc.h).  PyTorch is the
device-memory container and the stream provider; all arithmetic happens in the
hand-written gfx950 kernels.  Every function launches on the current torch
stream and does not synchronise.

Shapes use F frames, C views, P max people per view, N = C*P graph nodes.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _cabi
from ._cabi import MvmcSkeleton, check

_SEED_CACHE = {}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _req(t: torch.Tensor, dtype, name: str, shape=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError(f"{name}: expected a CUDA tensor")
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous tensor")
    if shape is not None:
        if t.dim() != len(shape) or any(s is not None and int(d) != int(s) for d, s in zip(t.shape, shape)):
            raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


# ----------------------------------------------------------------------------
# skeleton constants (inverse_kinematics.py:120-173) -- data, restated
# ----------------------------------------------------------------------------
SKEL_OFFSETS = np.array([
    [0, 0, 0], [0.15, 0, 0], [0, 0, -0.5], [0, 0, -0.5], [-0.15, 0, 0], [0, 0, -0.5],
    [0, 0, -0.5], [0, 0, 0.3], [0, 0, 0.3], [0.2, 0, 0], [0.3, 0, 0], [0.3, 0, 0],
    [-0.2, 0, 0], [-0.3, 0, 0], [-0.3, 0, 0], [0, -0.02, 0.15], [0.07, 0.02, 0.1],
    [-0.07, 0.02, 0.1]], dtype=np.float64)
SKEL_PARENTS = np.array([-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 9, 10, 8, 12, 13, 8, 15, 15], dtype=np.int32)
SKEL_SIDE_MAP = np.array([7, 0, 1, 2, 0, 1, 2, 8, 9, 3, 4, 5, 3, 4, 5, 10, 6, 6], dtype=np.int32)
SKEL_SIDE_JOINTS = np.array([1, 2, 3, 9, 10, 11, 16, 0, 7, 8, 15])


def skeleton_arrays():
    """(ref_bone_dirs[18,3], ref_side_bone_lens[11])."""
    lens = np.linalg.norm(SKEL_OFFSETS, axis=-1)
    dirs = SKEL_OFFSETS.copy()
    dirs[1:] = dirs[1:] / lens[1:, None]
    return dirs, lens[SKEL_SIDE_JOINTS].copy()


def make_skeleton(bone_dirs=None, parents=None, side_map=None, n_side=11, ref_side_lens=None) -> MvmcSkeleton:
    sk = MvmcSkeleton()
    rl = skeleton_arrays()[1] if ref_side_lens is None else np.asarray(ref_side_lens, np.float64)
    for k in range(18):
        sk.ref_side_lens[k] = float(rl[k]) if k < len(rl) else 0.0
    bd = skeleton_arrays()[0] if bone_dirs is None else np.asarray(bone_dirs, np.float64)
    pa = SKEL_PARENTS if parents is None else np.asarray(parents, np.int32)
    sm = SKEL_SIDE_MAP if side_map is None else np.asarray(side_map, np.int32)
    for j in range(18):
        for k in range(3):
            sk.bone_dirs[j][k] = float(bd[j, k])
        sk.parents[j] = int(pa[j])
        sk.side_map[j] = int(sm[j])
    sk.n_side = int(n_side)
    return sk


# ----------------------------------------------------------------------------
def als_seed_table(count: int, device) -> torch.Tensor:
    """First ``count`` doubles of RandomState(0).rand() (host MT19937 in the library), on ``device``."""
    key = (str(device), int(count))
    hit = _SEED_CACHE.get(key)
    if hit is not None:
        return hit
    buf = (C.c_double * count)()
    check(_cabi.load().mvmc_als_seed_table(C.cast(buf, C.c_void_p), count), "mvmc_als_seed_table")
    t = torch.from_numpy(np.frombuffer(buf, dtype=np.float64).copy()).to(device)
    _SEED_CACHE[key] = t
    return t


def ingest(kps: torch.Tensor, counts: Optional[torch.Tensor] = None, min_score=0.01, min_valid=4, min_bb=5.0):
    """IN-1/IN-2.  kps (F,C,P,25|17,3) f32|f64 -> (kps17 (F,C,P,17,3) f64, counts (F,C) i32)."""
    if kps.dtype not in (torch.float32, torch.float64):
        raise ValueError("ingest: kps must be float32 or float64")
    _req(kps, kps.dtype, "kps")
    if kps.dim() != 5 or kps.shape[3] not in (17, 25) or kps.shape[4] != 3:
        raise ValueError(f"ingest: expected (F,C,P,25|17,3), got {tuple(kps.shape)}")
    F, Cn, P, J, _ = kps.shape
    if counts is not None:
        _req(counts, torch.int32, "counts", (F, Cn))
    out = torch.empty((F, Cn, P, 17, 3), dtype=torch.float64, device=kps.device)
    cnt = torch.empty((F, Cn), dtype=torch.int32, device=kps.device)
    dt = _cabi.MVMC_F32 if kps.dtype == torch.float32 else _cabi.MVMC_F64
    check(_cabi.load().mvmc_ingest(_p(kps), dt, F, Cn, P, J, _p(counts), float(min_score), int(min_valid),
                                   float(min_bb), _p(out), _p(cnt), _stream()), "mvmc_ingest")
    return out, cnt


def fmats(K: torch.Tensor, Rt: torch.Tensor) -> torch.Tensor:
    """AS-1.  K (C,3,3), Rt (C,3,4) f64 -> F (C,C,3,3) f32."""
    Cn = K.shape[0]
    _req(K, torch.float64, "K", (Cn, 3, 3))
    _req(Rt, torch.float64, "Rt", (Cn, 3, 4))
    F = torch.empty((Cn, Cn, 3, 3), dtype=torch.float32, device=K.device)
    check(_cabi.load().mvmc_fmats(_p(K), _p(Rt), Cn, _p(F), _stream()), "mvmc_fmats")
    return F


def affinity(kps17: torch.Tensor, counts: torch.Tensor, Fm: torch.Tensor, want_D=True):
    """AS-2/AS-3.  -> (D, S) each (F,N,N) f32 in compact node order."""
    F, Cn, P = kps17.shape[:3]
    _req(kps17, torch.float64, "kps17", (F, Cn, P, 17, 3))
    _req(counts, torch.int32, "counts", (F, Cn))
    _req(Fm, torch.float32, "Fm", (Cn, Cn, 3, 3))
    N = Cn * P
    D = torch.empty((F, N, N), dtype=torch.float32, device=kps17.device) if want_D else None
    S = torch.empty((F, N, N), dtype=torch.float32, device=kps17.device)
    check(_cabi.load().mvmc_affinity(_p(kps17), _p(counts), _p(Fm), F, Cn, P, _p(D), _p(S), _stream()),
          "mvmc_affinity")
    return D, S


def als_associate(W: torch.Tensor, group_counts: torch.Tensor, g_max: int, want_mats=False):
    """AS-4/5/6.  W (F,N,N) f32|f64, group_counts (F,G) i32 -> dict(labels (F,N), n_clusters, iters[, x_bin, match_mat])."""
    if W.dtype not in (torch.float32, torch.float64):
        raise ValueError("als_associate: W must be float32 or float64")
    F, N, N2 = W.shape
    _req(W, W.dtype, "W", (F, N, N))
    G = group_counts.shape[1]
    _req(group_counts, torch.int32, "group_counts", (F, G))
    dev = W.device
    seed = als_seed_table(_cabi.MAX_NODES * _cabi.MAX_NODES, dev)
    labels = torch.empty((F, N), dtype=torch.int32, device=dev)
    ncl = torch.empty((F,), dtype=torch.int32, device=dev)
    iters = torch.empty((F,), dtype=torch.int32, device=dev)
    xb = torch.empty((F, N, N), dtype=torch.uint8, device=dev) if want_mats else None
    mm = torch.empty((F, N, N), dtype=torch.uint8, device=dev) if want_mats else None
    dt = _cabi.MVMC_F32 if W.dtype == torch.float32 else _cabi.MVMC_F64
    check(_cabi.load().mvmc_als_associate(_p(W), dt, _p(group_counts), F, G, N, int(g_max), _p(seed),
                                          seed.numel(), _p(xb), _p(mm), _p(labels), _p(ncl), _p(iters), _stream()),
          "mvmc_als_associate")
    return dict(labels=labels, n_clusters=ncl, iters=iters, x_bin=xb, match_mat=mm)


def svt_associate(S: torch.Tensor, group_counts: torch.Tensor, g_max: int, alpha=0.1, lam=50.0, mu=64.0, tol=5e-4, max_iter=20,
                  dual_stochastic=True, want_x=False):
    """match_svt (mv_association.py:321-411) + transform_closure + the cluster rule.  S (F,N,N) f32|f64, group_counts (F,G) i32 ->
    dict(x_bin, match_mat (F,N,N) u8, labels (F,N), n_clusters (F), iters (F)[, X (F,N,N) f64])."""
    if S.dtype not in (torch.float32, torch.float64):
        raise ValueError("svt_associate: S must be float32 or float64")
    F, N, _ = S.shape
    _req(S, S.dtype, "S", (F, N, N))
    G = group_counts.shape[1]
    _req(group_counts, torch.int32, "group_counts", (F, G))
    dev = S.device
    work = torch.empty((F, 3, N, N), dtype=torch.float64, device=dev)
    xb = torch.empty((F, N, N), dtype=torch.uint8, device=dev)
    xo = torch.empty((F, N, N), dtype=torch.float64, device=dev) if want_x else None
    iters = torch.empty((F,), dtype=torch.int32, device=dev)
    dt = _cabi.MVMC_F32 if S.dtype == torch.float32 else _cabi.MVMC_F64
    check(_cabi.load().mvmc_svt_associate(_p(S), dt, _p(group_counts), F, G, N, int(g_max), C.c_double(alpha), C.c_double(lam),
                                          C.c_double(mu), C.c_double(tol), int(max_iter), int(bool(dual_stochastic)), _p(work),
                                          _p(xb), _p(xo), _p(iters), _stream()), "mvmc_svt_associate")
    n_nodes = group_counts.sum(dim=1).to(torch.int32)
    mm, labels, ncl = closure_labels(xb, n_nodes)
    return dict(x_bin=xb, match_mat=mm, labels=labels, n_clusters=ncl, iters=iters, X=xo)


def closure_labels(x_bin: torch.Tensor, n_nodes: torch.Tensor, want_mat=True):
    """AS-5/AS-6 alone.  x_bin (F,N,N) u8, n_nodes (F) i32 -> (match_mat (F,N,N) u8 | None, labels (F,N), n_clusters (F))."""
    F, N, _ = x_bin.shape
    _req(x_bin, torch.uint8, "x_bin", (F, N, N))
    _req(n_nodes, torch.int32, "n_nodes", (F,))
    mm = torch.empty((F, N, N), dtype=torch.uint8, device=x_bin.device) if want_mat else None
    labels = torch.empty((F, N), dtype=torch.int32, device=x_bin.device)
    ncl = torch.empty((F,), dtype=torch.int32, device=x_bin.device)
    check(_cabi.load().mvmc_closure_labels(_p(x_bin), _p(n_nodes), F, N, _p(mm), _p(labels), _p(ncl), _stream()),
          "mvmc_closure_labels")
    return mm, labels, ncl


def cluster_members(labels: torch.Tensor, counts: torch.Tensor, p_max: int, k_max: int, v_max: int):
    """labels (F,N), counts (F,C) -> members (F,K,V) pose indices (-1 padded), n_members (F,K)."""
    F, Cn = counts.shape
    _req(counts, torch.int32, "counts")
    _req(labels, torch.int32, "labels", (F, Cn * p_max))
    mem = torch.empty((F, k_max, v_max), dtype=torch.int32, device=labels.device)
    nm = torch.empty((F, k_max), dtype=torch.int32, device=labels.device)
    check(_cabi.load().mvmc_cluster_members(_p(labels), _p(counts), F, Cn, p_max, k_max, v_max, _p(mem), _p(nm),
                                            _stream()), "mvmc_cluster_members")
    return mem, nm


def dlt(kps: torch.Tensor, Pmats: torch.Tensor, members: torch.Tensor, min_score=0.01, post_optimize=False) -> torch.Tensor:
    """TR-1/TR-2.  kps (F,C,P,J,3); Pmats (C,3,4); members (...,V) pose indices -> (...,J,4)."""
    F, Cn, P, J = kps.shape[:4]
    _req(kps, torch.float64, "kps", (F, Cn, P, J, 3))
    _req(Pmats, torch.float64, "Pmats", (Cn, 3, 4))
    _req(members, torch.int32, "members")
    mem = members.reshape(-1, members.shape[-1])
    B, V = mem.shape
    out = torch.empty((B, J, 4), dtype=torch.float64, device=kps.device)
    check(_cabi.load().mvmc_dlt(_p(kps), _p(Pmats), _p(mem), B, V, Cn, P, J, float(min_score), _p(out), _stream()),
          "mvmc_dlt")
    if post_optimize:
        check(_cabi.load().mvmc_triangulate_postopt(_p(kps), _p(Pmats), _p(mem), B, V, Cn, P, J, _p(out), _stream()),
              "mvmc_triangulate_postopt")
    return out.reshape(members.shape[:-1] + (J, 4))


def ingest_dlt(kps: torch.Tensor, counts: Optional[torch.Tensor], Pmats: torch.Tensor, members: torch.Tensor, min_score=0.01,
               ingest_min_score=0.01, min_valid=4, min_bb=5.0, want_counts=False, out_dtype=torch.float64):
    """ingest() + dlt() in one pass (include/mvmc.h: mvmc_ingest_dlt).  kps (F,C,P,25|17,3) f32|f64; members (F,K,V) i32 in ingest()'s
    output numbering, every cluster inside its own frame -> pts3d (F,K,17,4) f64 [, counts (F,C)].  out_dtype=torch.float32
    (float32 keypoints only; mvmc_ingest_dlt_f32): the same float64 arithmetic, the points rounded once at a 16-byte store."""
    if kps.dtype not in (torch.float32, torch.float64):
        raise ValueError("ingest_dlt: kps must be float32 or float64")
    _req(kps, kps.dtype, "kps")
    if kps.dim() != 5 or kps.shape[3] not in (17, 25) or kps.shape[4] != 3:
        raise ValueError(f"ingest_dlt: expected (F,C,P,25|17,3), got {tuple(kps.shape)}")
    F, Cn, P, J, _ = kps.shape
    if counts is not None:
        _req(counts, torch.int32, "counts", (F, Cn))
    _req(Pmats, torch.float64, "Pmats", (Cn, 3, 4))
    if members.dim() != 3 or members.shape[0] != F:
        raise ValueError("ingest_dlt: members must be (F,K,V)")
    _req(members, torch.int32, "members")
    K, V = members.shape[1:]
    if out_dtype not in (torch.float32, torch.float64) or (out_dtype == torch.float32 and kps.dtype != torch.float32):
        raise ValueError("ingest_dlt: out_dtype float32 needs float32 keypoints")
    out = torch.empty((F, K, 17, 4), dtype=out_dtype, device=kps.device)
    cnt = torch.empty((F, Cn), dtype=torch.int32, device=kps.device) if want_counts else None
    if out_dtype == torch.float32:
        check(_cabi.load().mvmc_ingest_dlt_f32(_p(kps), F, Cn, P, J, _p(counts), float(ingest_min_score), int(min_valid), float(min_bb),
                                               _p(Pmats), _p(members), K, V, float(min_score), _p(out), _p(cnt), _stream()),
              "mvmc_ingest_dlt_f32")
        return (out, cnt) if want_counts else out
    dt = _cabi.MVMC_F32 if kps.dtype == torch.float32 else _cabi.MVMC_F64
    check(_cabi.load().mvmc_ingest_dlt(_p(kps), dt, F, Cn, P, J, _p(counts), float(ingest_min_score), int(min_valid), float(min_bb),
                                       _p(Pmats), _p(members), K, V, float(min_score), _p(out), _p(cnt), _stream()), "mvmc_ingest_dlt")
    return (out, cnt) if want_counts else out


def fk(params: torch.Tensor, skeleton: Optional[MvmcSkeleton] = None, want_G=False):
    """FK-1/FK-2.  params (B, 57+n_side) f64 -> joints (B,18,3)[, G (B,18,4,4)]."""
    sk = skeleton if skeleton is not None else make_skeleton()
    B = params.shape[0]
    _req(params, torch.float64, "params", (B, 57 + sk.n_side))
    joints = torch.empty((B, 18, 3), dtype=torch.float64, device=params.device)
    G = torch.empty((B, 18, 4, 4), dtype=torch.float64, device=params.device) if want_G else None
    check(_cabi.load().mvmc_fk(C.byref(sk), _p(params), B, _p(joints), _p(G), _stream()), "mvmc_fk")
    return (joints, G) if want_G else joints


_IK_SCRATCH = {}


def _ik_scratch(n_problems: int, dev) -> torch.Tensor:
    """Workspace of the IK kernel's eigensolver fallback (include/mvmc.h: MVMC_IK_SCRATCH_DOUBLES per problem).
    One buffer per (device, stream), grown on demand and never read by the host."""
    key = (str(dev), torch.cuda.current_stream(dev).cuda_stream)
    buf = _IK_SCRATCH.get(key)
    if buf is None or buf.shape[0] < n_problems:
        buf = torch.empty((n_problems, _cabi.IK_SCRATCH_DOUBLES), dtype=torch.float64, device=dev)
        _IK_SCRATCH[key] = buf
    return buf


def ik_solve(kps17: torch.Tensor, Pmats: torch.Tensor, members: torch.Tensor,
             init_params: Optional[torch.Tensor] = None, cold: Optional[torch.Tensor] = None,
             max_nfev_cold=50, max_nfev_warm=5, skeleton: Optional[MvmcSkeleton] = None, want_info=True):
    """IK-1..IK-4.  members (B,V) -> params (B,68), joints (B,18,3), info (B,8)."""
    sk = skeleton if skeleton is not None else make_skeleton()
    F, Cn, P = kps17.shape[:3]
    _req(kps17, torch.float64, "kps17", (F, Cn, P, 17, 3))
    _req(Pmats, torch.float64, "Pmats", (Cn, 3, 4))
    _req(members, torch.int32, "members")
    B, V = members.shape
    dev = kps17.device
    if init_params is not None:
        _req(init_params, torch.float64, "init_params", (B, 68))
    if cold is not None:
        _req(cold, torch.uint8, "cold", (B,))
    if init_params is None and cold is not None:
        raise ValueError("ik_solve: warm problems need init_params")
    params = torch.empty((B, 68), dtype=torch.float64, device=dev)
    joints = torch.empty((B, 18, 3), dtype=torch.float64, device=dev)
    info = torch.empty((B, 8), dtype=torch.float64, device=dev) if want_info else None
    check(_cabi.load().mvmc_ik_solve(C.byref(sk), _p(kps17), _p(Pmats), _p(members), B, V, Cn, P, _p(init_params),
                                     _p(cold if init_params is not None else None), int(max_nfev_cold),
                                     int(max_nfev_warm), _p(params), _p(joints), _p(info), _p(_ik_scratch(B, dev)),
                                     _stream()),
          "mvmc_ik_solve")
    return params, joints, info


def ik_solve_stages(init_params: torch.Tensor, stage_mask: int, max_nfev: int, kps17: Optional[torch.Tensor] = None,
                    Pmats: Optional[torch.Tensor] = None, members: Optional[torch.Tensor] = None,
                    targets3d: Optional[torch.Tensor] = None, skeleton: Optional[MvmcSkeleton] = None):
    """Single stages of PoseSolver.solve (stage_mask 1 / 2 / 3) from init_params (B,68), either on the reprojection
    residual (kps17, Pmats, members as in ik_solve) or on 3-D targets (B,18,4) = x, y, z, weight per observation row."""
    sk = skeleton if skeleton is not None else make_skeleton()
    B = init_params.shape[0]
    _req(init_params, torch.float64, "init_params", (B, 68))
    dev = init_params.device
    if targets3d is not None:
        _req(targets3d, torch.float64, "targets3d", (B, 18, 4))
        V = Cn = P = 1
    else:
        if kps17 is None or Pmats is None or members is None:
            raise ValueError("ik_solve_stages: reprojection mode needs kps17, Pmats and members")
        F, Cn, P = kps17.shape[:3]
        _req(kps17, torch.float64, "kps17", (F, Cn, P, 17, 3))
        _req(Pmats, torch.float64, "Pmats", (Cn, 3, 4))
        _req(members, torch.int32, "members", (B, None))
        V = members.shape[1]
    params = torch.empty((B, 68), dtype=torch.float64, device=dev)
    joints = torch.empty((B, 18, 3), dtype=torch.float64, device=dev)
    info = torch.empty((B, 8), dtype=torch.float64, device=dev)
    check(_cabi.load().mvmc_ik_solve_stages(C.byref(sk), _p(kps17), _p(Pmats), _p(members), _p(targets3d), B, V, Cn, P,
                                            _p(init_params), int(stage_mask), int(max_nfev), _p(params), _p(joints),
                                            _p(info), _p(_ik_scratch(B, dev)), _stream()), "mvmc_ik_solve_stages")
    return params, joints, info


def ik_solve_fd(kps17: torch.Tensor, Pmats: torch.Tensor, members: torch.Tensor, init_params: Optional[torch.Tensor] = None,
                cold: Optional[torch.Tensor] = None, max_nfev_cold=50, max_nfev_warm=5, stage_mask=3,
                skeleton: Optional[MvmcSkeleton] = None):
    """Diagnostic: the same problems as ik_solve through the TRF-faithful solver (finite-difference Jacobian, SVD step;
    mvmc_debug_ik_solve_fd).  -> params (B,68), joints (B,18,3), info (B,8)."""
    sk = skeleton if skeleton is not None else make_skeleton()
    F, Cn, P = kps17.shape[:3]
    _req(kps17, torch.float64, "kps17", (F, Cn, P, 17, 3))
    _req(Pmats, torch.float64, "Pmats", (Cn, 3, 4))
    _req(members, torch.int32, "members")
    B, V = members.shape
    dev = kps17.device
    if init_params is not None:
        _req(init_params, torch.float64, "init_params", (B, 68))
    if cold is not None:
        _req(cold, torch.uint8, "cold", (B,))
    if init_params is None and cold is not None:
        raise ValueError("ik_solve_fd: warm problems need init_params")
    params = torch.empty((B, 68), dtype=torch.float64, device=dev)
    joints = torch.empty((B, 18, 3), dtype=torch.float64, device=dev)
    info = torch.empty((B, 8), dtype=torch.float64, device=dev)
    work = torch.empty((B, _cabi.IK_FD_WORK_DOUBLES), dtype=torch.float64, device=dev)
    check(_cabi.load().mvmc_debug_ik_solve_fd(C.byref(sk), _p(kps17), _p(Pmats), _p(members), B, V, Cn, P, _p(init_params),
                                              _p(cold if init_params is not None else None), int(max_nfev_cold),
                                              int(max_nfev_warm), int(stage_mask), _p(params), _p(joints), _p(info), _p(work),
                                              _stream()), "mvmc_debug_ik_solve_fd")
    return params, joints, info


def ik_model_step(kps17: torch.Tensor, Pmats: torch.Tensor, members: torch.Tensor, params: torch.Tensor, stage: int,
                  Delta: torch.Tensor, alpha0: torch.Tensor, skeleton: Optional[MvmcSkeleton] = None) -> torch.Tensor:
    """Diagnostic: one trust-region model + one trial step of the production IK from (params, Delta, alpha0) per problem
    (mvmc_debug_ik_model_step; layout of the (B, 240) result in include/mvmc.h)."""
    sk = skeleton if skeleton is not None else make_skeleton()
    F, Cn, P = kps17.shape[:3]
    _req(kps17, torch.float64, "kps17", (F, Cn, P, 17, 3))
    _req(Pmats, torch.float64, "Pmats", (Cn, 3, 4))
    _req(members, torch.int32, "members")
    B, V = members.shape
    _req(params, torch.float64, "params", (B, 68))
    _req(Delta, torch.float64, "Delta", (B,))
    _req(alpha0, torch.float64, "alpha0", (B,))
    dev = kps17.device
    out = torch.empty((B, _cabi.IK_STEP_OUT_DOUBLES), dtype=torch.float64, device=dev)
    check(_cabi.load().mvmc_debug_ik_model_step(C.byref(sk), _p(kps17), _p(Pmats), _p(members), B, V, Cn, P, _p(params), int(stage),
                                                _p(Delta), _p(alpha0), _p(out), _p(_ik_scratch(B, dev)), _stream()),
          "mvmc_debug_ik_model_step")
    return out


# ----------------------------------------------------------------------------
# temporal layer (match_spatial_time + tracker), batched over chains
# ----------------------------------------------------------------------------
def fmats_from_projections(Pmats: torch.Tensor) -> torch.Tensor:
    """AS-8 helper.  Pmats (C,3,4) f64 -> F2 (C,C,3,3) f64."""
    Cn = Pmats.shape[0]
    _req(Pmats, torch.float64, "Pmats", (Cn, 3, 4))
    F2 = torch.empty((Cn, Cn, 3, 3), dtype=torch.float64, device=Pmats.device)
    check(_cabi.load().mvmc_fmats_from_projections(_p(Pmats), Cn, _p(F2), _stream()), "mvmc_fmats_from_projections")
    return F2


def st_affinity(kps17, counts, frame_idx, track_joints, n_tracks, Pmats, F2, want_D=False, min_score=0.1):
    """AS-7/8/9.  -> (W (B,NS,NS) f64, D | None, group_counts (B,C+1) i32)."""
    F, Cn, P = kps17.shape[:3]
    B, T = track_joints.shape[:2]
    _req(kps17, torch.float64, "kps17", (F, Cn, P, 17, 3))
    _req(counts, torch.int32, "counts", (F, Cn))
    _req(frame_idx, torch.int32, "frame_idx", (B,))
    _req(track_joints, torch.float64, "track_joints", (B, T, 18, 3))
    _req(n_tracks, torch.int32, "n_tracks", (B,))
    _req(Pmats, torch.float64, "Pmats", (Cn, 3, 4))
    _req(F2, torch.float64, "F2", (Cn, Cn, 3, 3))
    NS = T + Cn * P
    W = torch.empty((B, NS, NS), dtype=torch.float64, device=kps17.device)
    D = torch.empty((B, NS, NS), dtype=torch.float64, device=kps17.device) if want_D else None
    gc = torch.empty((B, Cn + 1), dtype=torch.int32, device=kps17.device)
    check(_cabi.load().mvmc_st_affinity(_p(kps17), _p(counts), _p(frame_idx), _p(track_joints), _p(n_tracks), _p(Pmats),
                                        _p(F2), B, Cn, P, T, float(min_score), _p(W), _p(D), _p(gc), _stream()),
          "mvmc_st_affinity")
    return W, D, gc


def track_assign(labels_sp, ncl_sp, labels_st, ncl_st, counts, frame_idx, n_tracks, track_params, p_max, k_max, v_max, overflow=None):
    """TK-1 first half -> members (B,T+K,V), cold (B,T+K), init (B,T+K,68), status (B,T), n_new (B).
    overflow (B) i32 in/out: bit 0 set where a cluster or member did not fit (k_max new clusters, v_max views)."""
    B, T = track_params.shape[:2]
    Cn = counts.shape[1]
    dev_ = track_params.device
    _req(labels_sp, torch.int32, "labels_sp", (B, Cn * p_max))
    _req(labels_st, torch.int32, "labels_st", (B, T + Cn * p_max))
    _req(track_params, torch.float64, "track_params", (B, T, 68))
    NP = T + k_max
    mem = torch.empty((B, NP, v_max), dtype=torch.int32, device=dev_)
    cold = torch.empty((B, NP), dtype=torch.uint8, device=dev_)
    init = torch.empty((B, NP, 68), dtype=torch.float64, device=dev_)
    status = torch.empty((B, T), dtype=torch.int32, device=dev_)
    n_new = torch.empty((B,), dtype=torch.int32, device=dev_)
    check(_cabi.load().mvmc_track_assign(_p(labels_sp), _p(ncl_sp), _p(labels_st), _p(ncl_st), _p(counts), _p(frame_idx),
                                         _p(n_tracks), _p(track_params), B, Cn, p_max, T, k_max, v_max, _p(mem), _p(cold),
                                         _p(init), _p(status), _p(n_new), _p(overflow), _stream()), "mvmc_track_assign")
    return mem, cold, init, status, n_new


def track_commit(status, n_new, ik_params, ik_joints, track_params, track_joints, meta, n_tracks, next_id, n_dead,
                 k_max, n_inits=3, slot_src=None, overflow=None):
    """TK-1 second half: updates the tracklet table tensors in place (overflow bit 1: a new tracklet did not fit t_max slots)."""
    B, T = track_params.shape[:2]
    _req(ik_params, torch.float64, "ik_params", (B, T + k_max, 68))
    _req(ik_joints, torch.float64, "ik_joints", (B, T + k_max, 18, 3))
    _req(meta, torch.int32, "meta", (B, T, 4))
    check(_cabi.load().mvmc_track_commit(_p(status), _p(n_new), _p(ik_params), _p(ik_joints), B, T, k_max, n_inits,
                                         _p(track_params), _p(track_joints), _p(meta), _p(n_tracks), _p(next_id),
                                         _p(n_dead), _p(slot_src), _p(overflow), _stream()), "mvmc_track_commit")
