"""The IK kernel's eigensolver (Householder tridiagonalisation + Sturm multisection + twisted factorisation)
in isolation against numpy.linalg.eigh, on random PSD matrices shaped like J^T J: rank deficient, wide
dynamic range, optional repeated eigenvalues."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(A, g):
    from multiview_motion_capture_amd import _cabi
    lib = _cabi.load()
    d = torch.device("cuda:0")
    B, n, _ = A.shape
    At, gt = torch.from_numpy(A).to(d), torch.from_numpy(g).to(d)
    lam = torch.empty((B, n), dtype=torch.float64, device=d)
    Vt = torch.empty((B, n, n), dtype=torch.float64, device=d)
    k0 = torch.empty((B,), dtype=torch.int32, device=d)
    p = lambda t: C.c_void_p(t.data_ptr())
    cyc = torch.zeros((B, 5), dtype=torch.float64, device=d)
    st = lib.mvmc_debug_eigh(p(At), p(gt), B, n, p(lam), p(Vt), p(k0), p(cyc), None)
    assert st == 0
    torch.cuda.synchronize()
    _run.last_cycles = cyc.cpu().numpy()
    return lam.cpu().numpy(), Vt.cpu().numpy(), k0.cpu().numpy()


def _make(rng, n, rank, degenerate=False):
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    ev = np.zeros(n)
    ev[:rank] = 10.0 ** rng.uniform(-3, 8, size=rank)
    if degenerate:
        ev[1] = ev[0]
        ev[3] = ev[2] * (1 + 1e-12)
    A = (Q * ev) @ Q.T
    return 0.5 * (A + A.T)


@pytest.mark.parametrize("n,rank", [(50, 48), (40, 38), (50, 50), (12, 9)])
def test_eigh_against_numpy(n, rank):
    rng = np.random.default_rng(n * 100 + rank)
    B = 16
    A = np.array([_make(rng, n, rank) for _ in range(B)])
    g = rng.normal(size=(B, n))
    lam, Vt, k0 = _run(A, g)
    for b in range(B):
        w, V = np.linalg.eigh(A[b])
        lmax = w[-1]
        assert np.abs(lam[b][k0[b]:] - w[k0[b]:]).max() <= 1e-13 * lmax          # resolved eigenvalues
        assert k0[b] == (w <= 1e-13 * lmax).sum() or abs(k0[b] - (n - rank)) <= 1
        Vr = Vt[b][k0[b]:]                                                        # resolved eigenvectors (rows)
        # eigenvalues are accurate to eps*lam_max ABSOLUTE, so vectors of the smallest resolved eigenvalues
        # (down to 1e-11 lam_max here) are orthogonal to ~1e-8, not to rounding
        assert np.abs(Vr @ Vr.T - np.eye(len(Vr))).max() < 1e-8, "orthonormality"
        res = np.abs(A[b] @ Vr.T - Vr.T * lam[b][k0[b]:]).max()
        assert res <= 1e-11 * lmax, res
        assert not Vt[b][:k0[b]].any() and not lam[b][:k0[b]].any()  # null cluster: zero rows


def test_eigh_repeated_eigenvalues_still_give_the_right_projections():
    """Close / repeated non-null eigenvalues: individual vectors are not unique, the quantities the
    trust-region step uses (sum over the pair of (v.g) v / (lam + alpha)) must still be right."""
    rng = np.random.default_rng(5)
    n, B = 30, 8
    A = np.array([_make(rng, n, 27, degenerate=True) for _ in range(B)])
    g = rng.normal(size=(B, n))
    lam, Vt, k0 = _run(A, g)
    for b in range(B):
        w, V = np.linalg.eigh(A[b])
        alpha = 1e-3 * w[-1]
        kk = k0[b]
        ref = V[:, kk:] @ ((V[:, kk:].T @ g[b]) / (w[kk:] + alpha))
        got = Vt[b].T @ ((Vt[b] @ g[b]) / (lam[b] + alpha))
        err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        assert err < 1e-5, err


def test_eigh_on_real_ik_normal_matrices():
    """J^T J of actual Shelf IK problems (49 active columns + 1 zero pad, rank <= 48, exact zero row/column)."""
    import oracle_np as o
    import trf_np as t
    from conftest import load_golden
    g = load_golden("ik_cases.npz")
    warm = np.nonzero(~g["cold"])[0][:8]
    mats, grads = [], []
    for i in warm:
        v = int(g["n_views"][i])
        obs = np.array([o.add_mid_spine(p) for p in g["poses"][i, :v]])[:, o.IK_OBS_IDX, :]
        projs = np.asarray(g["projs"][i, :v])
        x = g["s2_x0"][i]
        J = t.ik_jacobian(x[:3], x[3:57], x[57:], obs, projs, True)
        f = o.ik_residual(x[:3], x[3:57], x[57:], obs, projs)
        act = np.nonzero(np.abs(J).max(axis=0) > 0)[0]
        Ja = np.zeros((J.shape[0], 50))
        Ja[:, :len(act)] = J[:, act]
        mats.append(Ja.T @ Ja)
        grads.append(Ja.T @ f)
    A, gr = np.array(mats), np.array(grads)
    lam, Vt, k0 = _run(A, gr)
    for b in range(len(warm)):
        w, V = np.linalg.eigh(A[b])
        lmax = w[-1]
        Vr = Vt[b][k0[b]:]
        assert np.abs(lam[b][k0[b]:] - w[k0[b]:]).max() <= 1e-12 * lmax
        assert np.abs(Vr @ Vr.T - np.eye(len(Vr))).max() < 1e-8
        assert np.abs(A[b] @ Vr.T - Vr.T * lam[b][k0[b]:]).max() <= 1e-11 * lmax
        for alpha in (1e-6 * lmax, 1e-2 * lmax):
            kk = k0[b]
            ref = V[:, kk:] @ ((V[:, kk:].T @ gr[b]) / (w[kk:] + alpha))
            got = Vt[b].T @ ((Vt[b] @ gr[b]) / (lam[b] + alpha))
            assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-8
    print("k0 on IK matrices:", k0)
    print("eigensolver phase cycles [tridiag, multisection, twisted, reorth, backtransform]:", _run.last_cycles.mean(0).round(0))
