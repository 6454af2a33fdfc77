"""Drop-in for the live matcher of the reference's mv_association.py (match_als, transform_closure)."""
from __future__ import annotations

import numpy as np
import torch

from . import device as dev
from .mv_math_util import _d


def match_als(W: np.ndarray, dimGroup, **kwargs):
    """mv_association.py:222-318 -> (match_mat (n,n) bool, X_bin (n,n) bool).
    W float32 (from geometry_affinity) or float64 (from match_spatial_time); W is not modified."""
    d = _d()
    W = np.asarray(W)
    if W.dtype not in (np.float32, np.float64):
        W = W.astype(np.float64)
    n = W.shape[0]
    cnt = np.diff(np.asarray(dimGroup)).astype(np.int32)
    if n == 0 or cnt.sum() != n:
        raise ValueError("match_als: dimGroup does not partition W")
    res = dev.als_associate(torch.as_tensor(np.ascontiguousarray(W[None]), device=d),
                            torch.as_tensor(cnt[None], device=d), g_max=int(cnt.max()), want_mats=True)
    if int(res["iters"][0]) < 0:
        raise ValueError("match_als: problem size outside the compiled kernel variants")
    return res["match_mat"][0].cpu().numpy().astype(bool), res["x_bin"][0].cpu().numpy().astype(bool)


def transform_closure(x_bin):
    """mv_association.py:99-121 (including its k = N-1 overwrite quirk) -> match_result_mat (n,n) bool."""
    d = _d()
    xb = np.ascontiguousarray(np.asarray(x_bin) != 0).astype(np.uint8)
    n = xb.shape[0]
    mm, _, _ = dev.closure_labels(torch.as_tensor(xb[None], device=d),
                                  torch.full((1,), n, dtype=torch.int32, device=d))
    return mm[0].cpu().numpy().astype(bool)
