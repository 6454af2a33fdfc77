"""GPU parity of the production IK solver against CONVERGED reference solves (tests/golden/ik_converged.npz).

The fixture holds the reference's own ``solve_pose_reproj`` + ``solve_pose_bone_lens_reproj`` (inverse_kinematics.py:202-277) run with an
evaluation budget large enough for SciPy to stop by ftol / xtol / gtol on both stages (oracle/gen_golden_ikconv.py), on Shelf clusters
of 2-5 views and on clusters of a 64-frame subset of synthetic config 4, cold and warm initial points.  A converged solve is a
minimum of the reprojection cost: well defined, unlike the 5-evaluation truncated solves (tests/test_gpu_ik.py), so the north star's
1e-4 is enforced here -- on the cost, on the joints the views observe, and on the rotations the observations determine.
"""
import numpy as np
import pytest
import torch

import oracle_np as o
from conftest import load_golden

pytestmark = pytest.mark.gpu


def _observed(poses, v, min_views, min_score=0.1):
    sc = np.array([o.add_mid_spine(p) for p in poses[:v]])[:, o.IK_OBS_IDX, 2]
    return o.IK_SKEL_IDX[(sc > min_score).sum(axis=0) >= min_views]


@pytest.fixture(scope="module")
def conv():
    from multiview_motion_capture_amd import device as dev
    g = load_golden("ik_converged.npz")
    d = torch.device("cuda:0")
    n, VP = g["poses"].shape[:2]
    # every (case, view slot) is its own "camera": kps17 (1, n*VP, 1, 17, 3), Pmats (n*VP, 3, 4), members index them directly
    kps = torch.from_numpy(np.ascontiguousarray(g["poses"].reshape(1, n * VP, 1, 17, 3))).to(d)
    Pm = np.ascontiguousarray(g["projs"].reshape(n * VP, 3, 4)).copy()
    Pm[np.abs(Pm).sum(axis=(1, 2)) == 0] = np.eye(3, 4)     # unused slots
    mem = -np.ones((n, VP), dtype=np.int32)
    for i in range(n):
        mem[i, :g["n_views"][i]] = i * VP + np.arange(g["n_views"][i])
    nfev = int(g["max_nfev"])
    p, j, info = dev.ik_solve_stages(torch.from_numpy(g["init"]).to(d), 3, nfev, kps17=kps, Pmats=torch.from_numpy(Pm).to(d),
                                     members=torch.from_numpy(mem).to(d))
    torch.cuda.synchronize()
    return dict(g=g, p=p.cpu().numpy(), j=j.cpu().numpy(), info=info.cpu().numpy())


def _rows(c):
    g, info = c["g"], c["info"]
    both = (g["s1_status"] > 0) & (g["s2_status"] > 0)
    dev_conv = (info[:, 2] > 0) & (info[:, 5] > 0)
    return both, dev_conv


def test_fixture_is_large_and_varied(conv):
    g = conv["g"]
    both, _ = _rows(conv)
    print("converged reference solves:", int(both.sum()), "of", len(both), "| by views:", np.bincount(g["n_views"][both]),
          "| shelf / synthetic:", int((both & (g["source"] == 0)).sum()), int((both & (g["source"] == 1)).sum()),
          "| warm initial points:", int((both & g["warm_init"]).sum()))
    assert both.sum() >= 200
    assert (both & (g["source"] == 0)).sum() >= 60 and (both & (g["source"] == 1)).sum() >= 60
    assert all(np.bincount(g["n_views"][both], minlength=6)[v] >= 10 for v in (2, 3, 4, 5))


def test_converged_solves_reach_the_reference_minimum(conv):
    """Cost within 1e-4 (relative) on every case the reference converged on, 2-view clusters included: the device never stops at a
    worse point.  (It may stop at a BETTER one: with two views the problem has several minima and the two trajectories can part.)"""
    g, info = conv["g"], conv["info"]
    both, dev_conv = _rows(conv)
    rel = (info[:, 3] - g["s2_cost"]) / g["s2_cost"]
    idx = np.nonzero(both)[0]
    print("relative cost difference (device - reference): median |.| %.2e, p99 %.2e, max %.2e, min %.2e; device converged on %d of %d" %
          (np.median(np.abs(rel[idx])), np.quantile(np.abs(rel[idx]), 0.99), rel[idx].max(), rel[idx].min(), int(dev_conv[idx].sum()), len(idx)))
    worse = idx[rel[idx] > 1e-4]
    for i in worse:
        print("  worse than the reference: case", i, "views", g["n_views"][i], "source", g["source"][i], "rel", rel[i], "status", info[i, [2, 5]],
              "nfev", info[i, [1, 4]], "ref nfev", g["s1_nfev"][i], g["s2_nfev"][i])
    assert len(worse) == 0


def test_joints_and_observable_rotations_within_1e4(conv):
    """On every case where both solvers are at the same minimum (cost equal to 1e-6; 2-view clusters included): every joint that
    >= 2 views see within 1e-4 of the scene scale, and the rotations the observations determine -- the root's, the upper spine's
    (three children) and the head's (two children) global rotation matrices -- within the position bar divided by the lever arm of the children (and 95 % of them within 1e-4 outright); the root's Euler angles
    directly where it is away from gimbal lock (|Ry| not within 0.05 of pi/2; SURVEY.md section 8d parity metric ii).
    One qualification, enforced per case: SciPy stops when a step reduces the cost by less than ftol = 1e-8 (relative).  Where the
    device ends MORE than 1e-8 below the reference's cost, the reference stopped short of the minimum, and along a weakly observed
    direction 1e-8 of cost is ~1e-4 of position: those cases (1 of 272 in the fixture) get 1e-3."""
    g, p, j, info = conv["g"], conv["p"], conv["j"], conv["info"]
    both, dev_conv = _rows(conv)
    bd, _ = o.skeleton_constants()
    rel = (info[:, 3] - g["s2_cost"]) / g["s2_cost"]
    sel = np.nonzero(both & dev_conv & (np.abs(rel) < 1e-6))[0]
    assert len(sel) >= 250, len(sel)
    djs, drs, des, worst_r, worst_e, n_rot, n_euler, n_short = [], [], [], 0.0, 0.0, 0, 0, 0
    for i in sel:
        v = int(g["n_views"][i])
        short = rel[i] < -1e-8          # the reference stopped short of the device's cost by more than its own ftol
        n_short += short
        tol = 1e-3 if short else 1e-4
        oj = _observed(g["poses"][i], v, 2)
        scale = np.abs(g["joints"][i]).max()
        dj = np.abs(j[i][oj] - g["joints"][i][oj]).max() / scale
        djs.append(dj)
        assert dj < tol, (i, v, dj, rel[i])
        xr = g["s2_x"][i]
        _, Gd = o.forward_kinematics(p[i, :3], p[i, 3:57], p[i, 57:], bd)
        _, Gr = o.forward_kinematics(xr[:3], xr[3:57], xr[57:], bd)
        seen = set(_observed(g["poses"][i], v, 2).tolist())
        # a rotation is observed through the positions of the joint's children, lever = their distance from the joint (hips 0.15 m,
        # shoulders 0.2 m, ears 0.12 m): the rotation bar is the position bar divided by the lever
        for jt, kids, lever in ((0, (1, 4, 7), 0.15), (8, (9, 12, 15), 0.15), (15, (16, 17), 0.12)):
            if all(k in seen for k in kids):
                dr = np.abs(Gd[jt][:3, :3] - Gr[jt][:3, :3]).max()
                drs.append(dr)
                worst_r = max(worst_r, dr * lever / scale)
                n_rot += 1
                assert dr < 3 * tol, (i, jt, dr)     # the rotation itself, no lever / scale (observed max 1.3e-4 over 659 rotations)
                assert dr * lever / scale < tol, (i, jt, dr)
        if all(k in seen for k in (1, 4, 7)) and abs(abs(xr[4]) - np.pi / 2) > 0.05:
            de = np.abs(np.angle(np.exp(1j * (p[i, 3:6] - xr[3:6])))).max()
            worst_e = max(worst_e, de)
            des.append(de)
            n_euler += 1
            assert de < 3 * tol, (i, de)             # radians, as they are (observed max 1.1e-4 over 216 cases)
            assert de * 0.15 / scale < tol, (i, de)
    djs = np.array(djs)
    print(f"{len(sel)} cases at the same minimum ({n_short} where the reference stopped short by more than its ftol): joint diff rel. to "
          f"scene scale median {np.median(djs):.2e} p99 {np.quantile(djs, 0.99):.2e} max {djs.max():.2e}; {(djs < 1e-4).sum()} within 1e-4; "
          f"rotation-matrix entry diff median {np.median(drs):.2e} p99 {np.quantile(drs, 0.99):.2e} max {max(drs):.2e} over {n_rot} "
          f"rotations (worst lever-normalised {worst_r:.2e}); root Euler angle diff median {np.median(des):.2e} p99 "
          f"{np.quantile(des, 0.99):.2e} max {worst_e:.2e} rad over {n_euler} cases")
    assert np.quantile(drs, 0.95) < 1e-4 and np.quantile(des, 0.95) < 1e-4
    assert (djs < 1e-4).mean() >= 0.99 and n_short <= 0.05 * len(sel)


def test_trf_faithful_solver_on_a_sample(conv):
    """The TRF-faithful device solver on a sample of the same cases: same minimum (cost 1e-4) -- it is the reference's method."""
    from multiview_motion_capture_amd import device as dev
    g = conv["g"]
    both, _ = _rows(conv)
    idx = np.nonzero(both & (g["n_views"] >= 3))[0][::6][:24]
    d = torch.device("cuda:0")
    VP = g["poses"].shape[1]
    kps = torch.from_numpy(np.ascontiguousarray(g["poses"][idx].reshape(1, len(idx) * VP, 1, 17, 3))).to(d)
    Pm = np.ascontiguousarray(g["projs"][idx].reshape(len(idx) * VP, 3, 4)).copy()
    Pm[np.abs(Pm).sum(axis=(1, 2)) == 0] = np.eye(3, 4)
    mem = -np.ones((len(idx), VP), dtype=np.int32)
    for b, i in enumerate(idx):
        mem[b, :g["n_views"][i]] = b * VP + np.arange(g["n_views"][i])
    nfev = int(g["max_nfev"])
    p, j, info = dev.ik_solve_fd(kps, torch.from_numpy(Pm).to(d), torch.from_numpy(mem).to(d), torch.from_numpy(g["init"][idx]).to(d),
                                 torch.zeros(len(idx), dtype=torch.uint8, device=d), nfev, nfev)
    torch.cuda.synchronize()
    info = info.cpu().numpy()
    rel = (info[:, 3] - g["s2_cost"][idx]) / g["s2_cost"][idx]
    print("TRF-faithful on", len(idx), "cases: rel cost median |.| %.2e max %.2e; status" % (np.median(np.abs(rel)), rel.max()),
          np.bincount(info[:, 5].astype(int)))
    assert (rel < 1e-4).all()
