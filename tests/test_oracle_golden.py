"""Pins the NumPy oracle (oracle/oracle_np.py) against golden vectors that were
captured from the reference itself (oracle/gen_golden.py).  CPU only."""
import numpy as np
import pytest

import oracle_np as o
from conftest import SPATIAL_FRAMES, load_golden


def test_fk_known_answers_bit_exact():
    """725 (param, joints) pairs shipped by the reference (traclets.pkl): max abs error 0."""
    g = load_golden("fk_known_answers.npz")
    err = 0.0
    for i in range(len(g["root"])):
        pos, _ = o.forward_kinematics(g["root"][i], g["euler"][i], g["blens_full"][i],
                                      bone_dirs=g["bone_dirs"], side_map=None)
        err = max(err, np.abs(pos - g["joints"][i]).max())
    assert err == 0.0
    assert np.array_equal(g["parents"], o.SKEL_PARENTS)


def test_skeleton_constants():
    dirs, side = o.skeleton_constants()
    g = load_golden("fk_known_answers.npz")
    assert np.array_equal(dirs, g["cur_bone_dirs"])
    assert np.array_equal(side, g["cur_side_lens"])
    assert np.array_equal(o.SIDE_TO_FULL, g["cur_side_map"])


def test_shelf_loader_and_filter(shelf_inputs, shelf_spatial):
    """IN-1/IN-2: gather 25->17 + filter_bad_pose reproduce the reference's filtered point sets."""
    for fi in SPATIAL_FRAMES:
        pts, sc, dim = [], [], [0]
        for c in range(5):
            k = 0
            for p in range(shelf_inputs["counts"][fi, c]):
                k17 = o.openpose25_to_coco17(shelf_inputs["kps25"][fi, c, p])
                if o.pose_is_good(k17):
                    pts.append(k17[:, :2])
                    sc.append(k17[:, 2:])
                    k += 1
            dim.append(dim[-1] + k)
        assert np.array_equal(np.array(dim), shelf_spatial[f"f{fi}_dim"])
        assert np.array_equal(np.array(pts), shelf_spatial[f"f{fi}_points"])
        assert np.array_equal(np.array(sc), shelf_spatial[f"f{fi}_scores"])


def test_calib_and_fmats(shelf_inputs, shelf_spatial):
    K, Rt = shelf_inputs["K"], shelf_inputs["Rt"]
    for c in range(5):
        P, _ = o.calib_from_k_rt(K[c], Rt[c])
        assert np.array_equal(P, shelf_inputs["P"][c])
    F = o.pairwise_f_mats(K, Rt)
    Fg = shelf_spatial["f1_F"]
    off = ~np.eye(5, dtype=bool)
    # f64 math -> f32 storage: allow 1 ulp (torch vs numpy matmul order)
    rel = np.abs(F[off] - Fg[off]) / np.abs(Fg[off]).max(axis=(1, 2), keepdims=True)
    assert rel.max() < 2e-7
    # F[i,i] is amplified rounding noise in the reference (T - R R^T T != 0) and is
    # never read by geometry_affinity (pairs a < b only): not compared.


@pytest.mark.parametrize("fi", SPATIAL_FRAMES)
def test_affinity_bit_exact(shelf_spatial, fi):
    g = shelf_spatial
    D, S = o.geometry_affinity(g[f"f{fi}_points"], g[f"f{fi}_F"], g[f"f{fi}_dim"].tolist())
    assert D.dtype == np.float32 and S.dtype == np.float32
    assert np.array_equal(D, g[f"f{fi}_D"])
    assert np.array_equal(S, g[f"f{fi}_S"])
    # the explicit pairwise-sum restatement equals numpy's f32 mean
    n = D.size
    assert np.float32(o.np_pairwise_sum_f32(D.ravel()) / np.float32(n)) == D.mean()


@pytest.mark.parametrize("fi", SPATIAL_FRAMES)
def test_als_closure_clusters(shelf_spatial, fi):
    g = shelf_spatial
    dim = g[f"f{fi}_dim"].tolist()
    mm, xb, iters = o.match_als(g[f"f{fi}_S"], dim, return_iters=True)
    assert np.array_equal(xb, g[f"f{fi}_x_bin"])
    assert np.array_equal(mm.astype(np.uint8), g[f"f{fi}_match_mat"])
    assert iters == int(g[f"f{fi}_als_iters"])
    clusters = o.parse_match_result(mm, len(mm), dim)
    assert len(clusters) == int(g[f"f{fi}_n_clusters"])
    for ci, cl in enumerate(clusters):
        assert np.array_equal(np.array(cl), g[f"f{fi}_cl{ci}"])
    lab = o.cluster_labels(mm, len(mm))
    for ci, cl in enumerate(clusters):
        assert all(lab[gi] == ci for _, _, gi in cl)


def _cluster_inputs(g, shelf_inputs, fi, ci):
    cl = g[f"f{fi}_cl{ci}"]
    pts, sc = g[f"f{fi}_points"], g[f"f{fi}_scores"]
    projs = np.array([shelf_inputs["P"][grp] for grp, _, _ in cl])
    grps = [np.concatenate([pts[gi], sc[gi]], axis=1) for _, _, gi in cl]
    return projs, grps


@pytest.mark.parametrize("fi", SPATIAL_FRAMES)
def test_dlt(shelf_spatial, shelf_inputs, fi):
    g = shelf_spatial
    for ci in range(int(g[f"f{fi}_n_clusters"])):
        if f"f{fi}_cl{ci}_dlt" not in g:
            continue
        projs, grps = _cluster_inputs(g, shelf_inputs, fi, ci)
        out = o.triangulate_groups(projs, grps, 0.01, False)
        assert np.array_equal(out, g[f"f{fi}_cl{ci}_dlt"])
        out = o.triangulate_groups(projs, grps, 0.01, True)
        ref = g[f"f{fi}_cl{ci}_dlt_post"]
        assert np.allclose(out, ref, rtol=1e-9, atol=1e-9)


def test_ik_cases_match_reference(ik_cases):
    """PoseSolver.solve() cold + warm: same SciPy, same residual -> identical iterates."""
    g = ik_cases
    n = len(g["frame"])
    idx = list(np.nonzero(g["cold"])[0][:2]) + list(np.nonzero(~g["cold"])[0][:6])
    for i in idx:
        v = int(g["n_views"][i])
        init = None if g["cold"][i] else (g["init_root"][i], g["init_euler"][i], g["init_blens"][i])
        (r, e, b), joints, info = o.pose_solver_solve(list(g["poses"][i, :v]), list(g["projs"][i, :v]), init,
                                                      return_info=True)
        assert info["res1"].nfev == g["s1_nfev"][i] and info["res2"].nfev == g["s2_nfev"][i]
        assert info["res1"].status == g["s1_status"][i] and info["res2"].status == g["s2_status"][i]
        assert np.allclose(info["res1"].x, g["s1_x"][i], rtol=0, atol=1e-7)
        assert np.allclose(np.concatenate([r, e.ravel(), b]), g["s2_x"][i], rtol=0, atol=1e-7)
        assert np.allclose(joints, g["joints"][i], rtol=0, atol=1e-7)
    assert n >= 8


def test_match_svt_oracle_against_reference_vectors():
    """mv_association.py:321-411 restated in NumPy against X_bin / match_mat / SVD call counts recorded from the reference itself
    (oracle/gen_golden_svt.py), float32 and float64 inputs."""
    g = load_golden("svt_cases.npz")
    names = sorted({k[:-2] for k in g.files if k.endswith("_S")})
    assert len(names) == 22
    for nm in names:
        mm, xb, info = o.match_svt(g[nm + "_S"], g[nm + "_dim"], return_info=True)
        assert np.array_equal(xb, g[nm + "_x_bin"]), nm
        assert np.array_equal(mm.astype(bool), g[nm + "_match_mat"]), nm
        assert min(info["iter"] + 1, 20) == int(g[nm + "_svd_calls"]), nm
