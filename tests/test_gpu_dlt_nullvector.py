"""The DLT's null vector as a property, independent of how well separated the smallest singular value is: the point the device returns,
as a unit 4-vector, must make ||A x|| equal the smallest singular value of the 2V x 4 system (mv_math_util.py:215-240: the last right
singular vector).  Covers the fast path (inverse iteration on L D L^T), the small-gap fallback (eigenvalue by Jacobi, then the iteration
shifted to it) with cameras a few millimetres apart and gross outliers, and both kernels (mvmc_dlt and the one-pass mvmc_ingest_dlt)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cams(rng, C, baseline):
    K = np.array([[1000.0, 0, 500], [0, 1000.0, 400], [0, 0, 1]])
    P = np.zeros((C, 3, 4))
    for c in range(C):
        R = np.eye(3)
        a = 0.3 * baseline * rng.standard_normal()
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        t = np.array([baseline * (c - (C - 1) / 2), 0.02 * baseline * rng.standard_normal(), 4.0])
        P[c] = K @ np.concatenate([R, t[:, None]], 1)
    return P


def _system(P, kp):
    rows = []
    for c in range(P.shape[0]):
        x, y = kp[c, 0], kp[c, 1]
        rows.append(x * P[c, 2] - P[c, 0])
        rows.append(y * P[c, 2] - P[c, 1])
    return np.array(rows)


@pytest.mark.parametrize("baseline,noise", [(1.0, 0.5), (0.002, 0.5), (1.0, 80.0), (1e-5, 2.0)])
def test_null_vector_reaches_the_smallest_singular_value(baseline, noise):
    from multiview_motion_capture_amd import device as dev
    rng = np.random.default_rng(20260301)
    C, F = 5, 64
    P = _cams(rng, C, baseline)
    X = np.concatenate([rng.uniform(-1, 1, (F, 17, 2)), rng.uniform(-0.5, 0.5, (F, 17, 1))], 2)
    kps = np.zeros((F, C, 1, 17, 3))
    for c in range(C):
        h = np.einsum("ij,fkj->fki", P[c], np.concatenate([X, np.ones((F, 17, 1))], 2))
        kps[:, c, 0, :, :2] = h[..., :2] / h[..., 2:3] + noise * rng.standard_normal((F, 17, 2))
        kps[:, c, 0, :, 2] = 0.9
    members = (np.arange(F)[:, None] * C + np.arange(C)[None, :]).astype(np.int32)          # pose index (f C + c) P + 0, P = 1
    kd, Pd, md = torch.from_numpy(kps).cuda(), torch.from_numpy(P).cuda(), torch.from_numpy(members).cuda()
    out = dev.dlt(kd, Pd, md).cpu().numpy().reshape(F, 17, 4)
    fused = dev.ingest_dlt(kd, None, Pd, md.view(F, 1, C), ingest_min_score=-1.0, min_valid=0, min_bb=-1.0).cpu().numpy().reshape(F, 17, 4)
    assert np.array_equal(out, fused)                        # the two kernels share dlt_point: bit for bit
    worst = 0.0
    for f in range(F):
        for j in range(17):
            A = _system(P, kps[f, :, 0, j])
            s = np.linalg.svd(A, compute_uv=False)
            x = np.append(out[f, j, :3], 1.0)
            assert np.isfinite(x).all()
            res = np.linalg.norm(A @ (x / np.linalg.norm(x)))
            # ||A x|| >= s_min for every unit x; the returned one must reach it (relative to the system's scale s_max)
            worst = max(worst, (res - s[-1]) / s[0])
    assert worst < 1e-9, worst


def test_a_point_without_a_unique_null_vector_is_nan_not_the_origin():
    """A cluster of ONE view (HotPath.triangulate runs the DLT on every cluster) and two views on the same line of sight: the 2V x 4
    system has a null space of more than one dimension, the reference returns whichever vector LAPACK produces.  The device says so
    (NaN) instead of dressing the iteration's start vector up as the point (0, 0, 0); the mean score is still the reference's."""
    from multiview_motion_capture_amd import device as dev
    rng = np.random.default_rng(7)
    C = 3
    P = _cams(rng, C, 1.0)
    P[2] = P[1]                                               # views 1 and 2: the same camera
    kps = np.zeros((2, C, 1, 17, 3))
    kps[..., :2] = rng.uniform(100, 800, (2, C, 1, 17, 2))
    kps[:, 2] = kps[:, 1]                                     # ... seeing the same pixels
    kps[..., 2] = 0.8
    members = np.array([[0, -1, -1], [4, 5, -1]], dtype=np.int32)    # frame 0: view 0 alone; frame 1: the two coincident views
    out = dev.dlt(torch.from_numpy(kps).cuda(), torch.from_numpy(P).cuda(), torch.from_numpy(members).cuda()).cpu().numpy().reshape(2, 17, 4)
    assert np.isnan(out[..., :3]).all(), out[:, :2]
    assert np.allclose(out[..., 3], 0.8)
