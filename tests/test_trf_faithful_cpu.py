"""CPU: the literal C++ restatement of the reference's least-squares solver (csrc/mvmc_trf_faithful.h -- SciPy's TRF with 2-point
finite differences and an SVD trust-region step, as called at inverse_kinematics.py:236,274 and mv_math_util.py:203), built for the
host (oracle/cpu_twin), against the reference's golden IK solves.  The same header compiles for the device as the TRF-faithful
diagnostic solver (tests/test_gpu_ik.py, tests/test_gpu_ik_converged.py) and is the IK of the C++ CPU baseline."""
import ctypes

import numpy as np
from scipy.optimize import least_squares

import oracle_np as o
from conftest import load_golden
from helpers import cpu_twin, twin_postopt_root


def _solve(lib, g, i, nfev):
    bd, _ = o.skeleton_constants()
    v = int(g["n_views"][i])
    pose18 = np.ascontiguousarray(np.array([o.add_mid_spine(p) for p in g["poses"][i, :v]]))
    Pm = np.ascontiguousarray(g["projs"][i, :v])
    init = g["init"][i] if "init" in g.files else np.concatenate([g["s1_x0"][i], g["init_blens"][i]])
    bl = np.ascontiguousarray(init[57:])
    par = np.array(o.SKEL_PARENTS, dtype=np.int32)
    smap = np.array(o.SIDE_TO_FULL, dtype=np.int32)
    bdc = np.ascontiguousarray(bd)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    x, out1, out2 = np.ascontiguousarray(init.copy()), np.zeros(4), np.zeros(4)
    lib.trf_check_ik(p(bdc), p(par), p(smap), 11, p(pose18), p(Pm), v, p(bl), 0, nfev, p(x), p(out1))
    lib.trf_check_ik(p(bdc), p(par), p(smap), 11, p(pose18), p(Pm), v, p(bl), 1, nfev, p(x), p(out2))
    pos, _ = o.forward_kinematics(x[:3], x[3:57], x[57:], bd)
    return x, pos, out1, out2


def test_twin_reaches_the_reference_minimum_on_converged_solves():
    lib = cpu_twin()
    g = load_golden("ik_converged.npz")
    both = np.nonzero((g["s1_status"] > 0) & (g["s2_status"] > 0) & (g["n_views"] >= 3))[0][::5]
    rel = []
    for i in both:
        x, pos, o1, o2 = _solve(lib, g, i, int(g["max_nfev"]))
        assert o1[2] > 0 and o2[2] > 0, (i, o1, o2)
        rel.append((o2[0] - g["s2_cost"][i]) / g["s2_cost"][i])
    rel = np.array(rel)
    print(f"{len(both)} converged cases: rel cost diff median |.| {np.median(np.abs(rel)):.2e} max {rel.max():.2e} min {rel.min():.2e}")
    assert (rel < 1e-6).all() and np.median(np.abs(rel)) < 1e-8


def test_twin_postopt_follows_scipy_where_the_step_is_accepted():
    """ik_cases case 7: a 2-view cluster on which the reference ACCEPTS its post-optimisation step (a rank-deficient minimum-norm
    Gauss-Newton step stretched to |p| = |x0|: the hips move 0.6 m).  The twin lands within 1e-3 of that move; SciPy itself moves by
    up to 1e-5 of it when its start point changes by one ulp (the share its LAPACK noise triplets take)."""
    g = load_golden("ik_cases.npz")
    i = 7
    v = int(g["n_views"][i])
    poses18 = [o.add_mid_spine(q) for q in g["poses"][i, :v]]
    projs = g["projs"][i, :v]
    ref = o.triangulate_groups(projs, poses18, 0.01, True)
    dlt = o.triangulate_groups(projs, poses18, 0.01, False)
    root = 0.5 * (ref[11, :3] + ref[12, :3])
    move = np.abs(root - 0.5 * (dlt[11, :3] + dlt[12, :3])).max()
    assert move > 0.1
    assert np.abs(twin_postopt_root(projs, poses18) - root).max() < 1e-3 * move
    # SciPy under 1-ulp perturbations of the start point
    groups = [np.asarray(p) for p in poses18]

    def res(x):
        X = x.reshape((-1, 3))
        homo = np.concatenate([X, np.ones((X.shape[0], 1))], axis=-1).T
        d = []
        for vv in range(len(projs)):
            h = projs[vv] @ homo
            uv = (h[:2] / (h[2] + 1e-6)).T
            d.append(np.linalg.norm(uv - groups[vv][:, :2], axis=-1) * groups[vv][:, -1])
        return np.array(d).flatten()

    rng = np.random.default_rng(0)
    x0 = dlt[:, :3].ravel()
    spread = []
    for _ in range(4):
        r = least_squares(res, x0 * (1 + rng.choice([-1, 0, 1], size=x0.shape) * 2.2e-16), max_nfev=2)
        X = r.x.reshape(18, 3)
        spread.append(np.abs(0.5 * (X[11] + X[12]) - root).max())
    print("SciPy's own root under 1-ulp changes of the start point moves by", max(spread), "m; accepted move", move)
    assert 1e-7 < max(spread) < 1e-3 * move
