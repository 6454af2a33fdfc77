"""Evidence for the IK parity gates (DESIGN.md "IK parity"): the reference's IK output is not
reproducible to 1e-4 even by itself.  The same SciPy solver on the same inputs, with the
projection product of the residual written as one einsum instead of per-view matmuls (identical
mathematics, different floating-point summation order), lands 1e-4 .. 1e-2 m away in the joints
and up to a few % in cost, because the solve is truncated at max_nfev = 5 and the trust-region
step is normalised to |p| = Delta along near-null directions filled with finite-difference noise."""
import numpy as np
from scipy.optimize import least_squares

import oracle_np as o
from conftest import load_golden


def _residual_einsum(root, euler, blens, obs, projs, bd):
    pos, _ = o.forward_kinematics(root, euler, blens, bd)
    X = pos[o.IK_SKEL_IDX]
    h = np.einsum('vik,jk->vji', projs, np.concatenate([X, np.ones((len(X), 1))], axis=1))
    uv = h[..., :2] / (1e-5 + h[..., 2:3])
    return ((uv - obs[..., :2]) * obs[..., 2:3]).ravel()


def test_reference_ik_moves_under_float_equivalent_reformulation():
    g = load_golden("ik_cases.npz")
    bd, _ = o.skeleton_constants()
    warm = np.nonzero(~g["cold"])[0][:16]
    dj, rc = [], []
    for i in warm:
        v = int(g["n_views"][i])
        obs = np.array([o.add_mid_spine(p) for p in g["poses"][i, :v]])[:, o.IK_OBS_IDX, :]
        projs = np.asarray(g["projs"][i, :v])
        bl = g["init_blens"][i]
        r1 = least_squares(lambda x: _residual_einsum(x[:3], x[3:].reshape(-1, 3), bl, obs, projs, bd),
                           g["s1_x0"][i], max_nfev=5)
        r2 = least_squares(lambda x: _residual_einsum(x[:3], x[3:57].reshape(-1, 3), x[57:], obs, projs, bd),
                           np.concatenate([r1.x, bl]), max_nfev=5)
        pos, _ = o.forward_kinematics(r2.x[:3], r2.x[3:57], r2.x[57:], bd)
        dj.append(np.abs(pos - g["joints"][i]).max())
        rc.append(abs(r2.cost - g["s2_cost"][i]) / g["s2_cost"][i])
    dj, rc = np.array(dj), np.array(rc)
    print("reference vs float-equivalent reference: joint diff median %.2e max %.2e; rel cost max %.2e"
          % (np.median(dj), dj.max(), rc.max()))
    # the reference disagrees with itself by far more than the 1e-4 the north star asks for
    assert np.median(dj) > 1e-4
    assert dj.max() > 1e-3


def test_band_of_float_equivalent_twins_on_all_truncated_warm_cases():
    """The band the GPU gate uses (tests/test_gpu_ik.py), computed here on all 45 truncated warm cases: independent implementations
    of the reference's OWN algorithm -- SciPy on an einsum-formulated residual, and the C++ restatement of SciPy's TRF
    (csrc/mvmc_trf_faithful.h built for the host) -- land 2e-3 .. 5e-3 m (median) from the reference's joints."""
    import ctypes
    from helpers import cpu_twin
    g = load_golden("ik_cases.npz")
    bd, _ = o.skeleton_constants()
    lib = cpu_twin()
    par = np.array(o.SKEL_PARENTS, dtype=np.int32)
    smap = np.array(o.SIDE_TO_FULL, dtype=np.int32)
    bdc = np.ascontiguousarray(bd)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    warm = np.nonzero(~g["cold"])[0]
    d_einsum, d_cpp = [], []
    for i in warm:
        v = int(g["n_views"][i])
        obs = np.array([o.add_mid_spine(q) for q in g["poses"][i, :v]])[:, o.IK_OBS_IDX, :]
        projs = np.asarray(g["projs"][i, :v])
        bl = g["init_blens"][i]
        r1 = least_squares(lambda x: _residual_einsum(x[:3], x[3:].reshape(-1, 3), bl, obs, projs, bd), g["s1_x0"][i], max_nfev=5)
        r2 = least_squares(lambda x: _residual_einsum(x[:3], x[3:57].reshape(-1, 3), x[57:], obs, projs, bd),
                           np.concatenate([r1.x, bl]), max_nfev=5)
        pos, _ = o.forward_kinematics(r2.x[:3], r2.x[3:57], r2.x[57:], bd)
        d_einsum.append(np.abs(pos - g["joints"][i]).max())
        pose18 = np.ascontiguousarray(np.array([o.add_mid_spine(q) for q in g["poses"][i, :v]]))
        Pm = np.ascontiguousarray(projs)
        x = np.ascontiguousarray(np.concatenate([g["s1_x0"][i], bl]))
        blc, out = np.ascontiguousarray(bl), np.zeros(4)
        lib.trf_check_ik(p(bdc), p(par), p(smap), 11, p(pose18), p(Pm), v, p(blc), 0, 5, p(x), p(out))
        lib.trf_check_ik(p(bdc), p(par), p(smap), 11, p(pose18), p(Pm), v, p(blc), 1, 5, p(x), p(out))
        pos, _ = o.forward_kinematics(x[:3], x[3:57], x[57:], bd)
        d_cpp.append(np.abs(pos - g["joints"][i]).max())
    d_einsum, d_cpp = np.array(d_einsum), np.array(d_cpp)
    print("45 truncated warm solves, max joint distance to the reference: SciPy/einsum median %.2e p90 %.2e | C++ TRF restatement "
          "median %.2e p90 %.2e" % (np.median(d_einsum), np.quantile(d_einsum, 0.9), np.median(d_cpp), np.quantile(d_cpp, 0.9)))
    for d in (d_einsum, d_cpp):
        assert 1e-3 < np.median(d) < 1e-2 and np.quantile(d, 0.9) > 5e-3
