"""Drop-in for the hot-path functions of the reference's mv_math_util.py (same names, argument
order and return types; numpy by value).  Every function uploads its operands, runs the gfx950
kernel behind include/mvmc.h and downloads the result -- there is no CPU arithmetic path."""
from __future__ import annotations

from typing import List

import numpy as np
import torch

from . import device as dev
from .common import Calib


def _d():
    if not torch.cuda.is_available():
        raise RuntimeError("multiview_motion_capture_amd needs an MI355X (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def calc_pairwise_f_mats(calibs: List[Calib]) -> np.ndarray:
    """mv_math_util.py:267-285 -> (C,C,3,3) float32."""
    d = _d()
    K = torch.as_tensor(np.array([np.asarray(c.K, np.float64) for c in calibs]), device=d)
    Rt = torch.as_tensor(np.array([np.asarray(c.Rt, np.float64) for c in calibs]), device=d)
    return dev.fmats(K.contiguous(), Rt.contiguous()).cpu().numpy()


def _pack_nodes(points_set, dimGroup, scores=None):
    """(M,J,2) + prefix sums -> kps (1,C,P,J,3) f64, counts (1,C) i32."""
    pts = np.asarray(points_set, np.float64)
    cnt = np.diff(np.asarray(dimGroup)).astype(np.int32)
    C, P, J = len(cnt), max(int(cnt.max()), 1), pts.shape[1]
    kps = np.zeros((1, C, P, J, 3))
    for c in range(C):
        lo = int(dimGroup[c])
        kps[0, c, :cnt[c], :, :2] = pts[lo:lo + cnt[c]]
        kps[0, c, :cnt[c], :, 2] = 1.0 if scores is None else np.asarray(scores, np.float64).reshape(len(pts), J)[lo:lo + cnt[c]]
    return kps, cnt[None]


def geometry_affinity(points_set, Fs, dimGroup):
    """mv_math_util.py:320-351 -> (distance (M,M) f32, affinity (M,M) f32)."""
    d = _d()
    M = len(points_set)
    if np.asarray(points_set).shape[1] != 17:
        raise ValueError("geometry_affinity: 17 COCO joints per pose expected (mv_math_util.py:308)")
    kps, cnt = _pack_nodes(points_set, dimGroup)
    D, S = dev.affinity(torch.as_tensor(kps, device=d), torch.as_tensor(cnt, device=d),
                        torch.as_tensor(np.ascontiguousarray(Fs, np.float32), device=d))
    # compact node order -> the caller's order (identical: nodes are grouped by view already)
    C, P = kps.shape[1:3]
    return D[0, :M, :M].cpu().numpy(), S[0, :M, :M].cpu().numpy()


def triangulate_point_groups_from_multiple_views_linear(proj_matricies, points_grps, min_score,
                                                        post_optimize=False, n_max_iter=2):
    """mv_math_util.py:152-212.  proj_matricies V x (3,4); points_grps V x (J,3) -> (J,4)."""
    if post_optimize and n_max_iter != 2:
        raise ValueError("post_optimize: only n_max_iter = 2 (the reference's default, one trial step) is built")
    d = _d()
    V = len(points_grps)
    J = len(points_grps[0])
    kps = np.zeros((1, V, 1, J, 3))
    for v in range(V):
        kps[0, v, 0] = np.asarray(points_grps[v], np.float64)
    P = torch.as_tensor(np.array([np.asarray(p, np.float64) for p in proj_matricies]), device=d).contiguous()
    mem = torch.arange(V, dtype=torch.int32, device=d)[None]
    return dev.dlt(torch.as_tensor(kps, device=d), P, mem, float(min_score), post_optimize=bool(post_optimize))[0].cpu().numpy()


def triangulate_point_from_multiple_views_linear(proj_matricies, points):
    """mv_math_util.py:215-240.  (N,3,4), (N,2) -> (3,)."""
    pts = np.concatenate([np.asarray(points, np.float64), np.ones((len(points), 1))], axis=1)
    out = triangulate_point_groups_from_multiple_views_linear(proj_matricies, [p[None] for p in pts], 0.0)
    return out[0, :3]


def get_fundamental_matrix(p1, p2) -> np.ndarray:
    """mv_math_util.py:57-77: F with x2^T F x1 = 0 from two 3x4 projection matrices."""
    d = _d()
    P = torch.as_tensor(np.array([np.asarray(p1, np.float64), np.asarray(p2, np.float64)]), device=d).contiguous()
    return dev.fmats_from_projections(P)[0, 1].cpu().numpy()


def _pair_graph(joints3d, views, projs, min_score):
    """One-frame match_spatial_time graph with optional tracklet 0 and one pose per view -> raw D (numpy)."""
    d = _d()
    C = len(views)
    kps = torch.as_tensor(np.array([np.asarray(v, np.float64) for v in views])[None, :, None], device=d).contiguous()
    cnt = torch.ones((1, C), dtype=torch.int32, device=d)
    P = torch.as_tensor(np.array([np.asarray(p, np.float64) for p in projs]), device=d).contiguous()
    tj = torch.zeros((1, 1, 18, 3), dtype=torch.float64, device=d)
    if joints3d is not None:
        tj[0, 0] = torch.as_tensor(np.asarray(joints3d, np.float64), device=d)
    _, D, _ = dev.st_affinity(kps, cnt, torch.zeros(1, dtype=torch.int32, device=d), tj,
                              torch.ones(1, dtype=torch.int32, device=d), P, dev.fmats_from_projections(P),
                              want_D=True, min_score=min_score)
    return D[0].cpu().numpy()


def calc_epipolar_error(cam1: Calib, keypoints_1, scores_1, cam2: Calib, keypoints_2, scores_2,
                        min_valid_kps_score=0.05, invalid_default_error=np.nan):
    """mv_math_util.py:80-115 (17 COCO joints)."""
    if len(keypoints_1) == 0:
        return invalid_default_error
    k1 = np.concatenate([np.asarray(keypoints_1, np.float64), np.asarray(scores_1, np.float64).reshape(-1, 1)], axis=1)
    k2 = np.concatenate([np.asarray(keypoints_2, np.float64), np.asarray(scores_2, np.float64).reshape(-1, 1)], axis=1)
    e = float(_pair_graph(None, [k1, k2], [cam1.P, cam2.P], min_valid_kps_score)[1, 2])
    return invalid_default_error if np.isnan(e) else e
