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


def match_svt(S, dimGroup, **kwargs):
    """mv_association.py:321-411 -> (match_mat (n,n) bool, X_bin (n,n) bool).  Keyword arguments as in the reference (alpha, tol,
    maxIter, _lambda, mu, dual_stochastic_SVT; verbose is accepted and ignored); pselect must be 1 and eigenvalues False.  S is not
    modified (the reference zeroes the diagonal of the caller's array)."""
    if kwargs.get("pselect", 1) != 1 or kwargs.get("eigenvalues", False):
        raise ValueError("match_svt: only pselect=1, eigenvalues=False are implemented")
    d = _d()
    S = np.asarray(S)
    if S.dtype not in (np.float32, np.float64):
        S = S.astype(np.float64)
    n = S.shape[0]
    cnt = np.diff(np.asarray(dimGroup)).astype(np.int32)
    if n == 0 or cnt.sum() != n:
        raise ValueError("match_svt: dimGroup does not partition S")
    res = dev.svt_associate(torch.as_tensor(np.ascontiguousarray(S[None]), device=d), torch.as_tensor(cnt[None], device=d),
                            g_max=int(cnt.max()), alpha=kwargs.get("alpha", 0.1), lam=kwargs.get("_lambda", 50),
                            mu=kwargs.get("mu", 64), tol=kwargs.get("tol", 5e-4), max_iter=kwargs.get("maxIter", 20),
                            dual_stochastic=kwargs.get("dual_stochastic_SVT", True))
    return res["match_mat"][0].cpu().numpy().astype(bool), res["x_bin"][0].cpu().numpy().astype(bool)


def transform_closure(x_bin):
    """mv_association.py:99-121 (including its k = N-1 overwrite quirk) -> match_result_mat (n,n) bool."""
    d = _d()
    xb = np.ascontiguousarray(np.asarray(x_bin) != 0).astype(np.uint8)
    n = xb.shape[0]
    mm, _, _ = dev.closure_labels(torch.as_tensor(xb[None], device=d),
                                  torch.full((1,), n, dtype=torch.int32, device=d))
    return mm[0].cpu().numpy().astype(bool)
