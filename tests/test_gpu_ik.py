"""GPU parity of the IK solver (mvmc_ik_solve) -- see DESIGN.md "IK parity".

The reference's IK (SciPy TRF, 2-point finite differences, truncated at max_nfev) is numerically
chaotic: tests/test_ik_sensitivity.py shows that a float-equivalent re-ordering of one matmul inside
the residual moves its own answer by 1e-3..1e-2 m.  The gates are therefore layered:
  1. max_nfev = 1 (no step): FK + residual cost identical to the oracle            (rel 1e-12)
  2. one and two trust-region steps with an active constraint: cost / joints vs the CPU restatement
     of the device algorithm (oracle/trf_np.py: analytic Jacobian, normal equations)  (1e-7)
  3. converged cold starts with >= 3 views: joints and cost vs the REFERENCE golden  (1e-4 rel)
  4. truncated warm solves: inside the reference's own rounding-sensitivity band.
"""
import numpy as np
import pytest
import torch

import oracle_np as o
import trf_np as t
from conftest import load_golden

pytestmark = pytest.mark.gpu


def _pack(g, idx, Pshelf):
    """ik_cases rows -> (kps17 (B,5,1,17,3), members (B,5), init (B,68), cold (B,))."""
    B = len(idx)
    kps = np.zeros((B, 5, 1, 17, 3))
    mem = -np.ones((B, 5), dtype=np.int32)
    init = np.zeros((B, 68))
    cold = np.zeros(B, dtype=np.uint8)
    for b, i in enumerate(idx):
        v = int(g["n_views"][i])
        for s in range(v):
            cam = int(np.argmin([np.abs(Pshelf[c] - g["projs"][i, s]).max() for c in range(5)]))
            assert np.array_equal(Pshelf[cam], g["projs"][i, s])
            assert not kps[b, cam].any(), "duplicate view in a case is not representable in this packing"
            kps[b, cam, 0] = g["poses"][i, s]
            mem[b, s] = b * 5 + cam
        cold[b] = g["cold"][i]
        init[b] = np.concatenate([g["init_root"][i], g["init_euler"][i].ravel(), g["init_blens"][i]])
    return kps, mem, init, cold


def _usable(g):
    """cases without two poses from the same camera (the packing above needs distinct cameras)."""
    Pshelf = load_golden("shelf_inputs.npz")["P"]
    ok = []
    for i in range(len(g["frame"])):
        cams = [int(np.argmin([np.abs(Pshelf[c] - g["projs"][i, s]).max() for c in range(5)]))
                for s in range(int(g["n_views"][i]))]
        if len(set(cams)) == len(cams):
            ok.append(i)
    return np.array(ok), Pshelf


@pytest.fixture(scope="module")
def ctx():
    from multiview_motion_capture_amd import device as dev
    g = load_golden("ik_cases.npz")
    idx, Pshelf = _usable(g)
    return dict(dev=dev, g=g, idx=idx, P=Pshelf, d=torch.device("cuda:0"))


def _run(ctx, idx, nfev_cold, nfev_warm, force_init=None):
    dev, d = ctx["dev"], ctx["d"]
    kps, mem, init, cold = _pack(ctx["g"], idx, ctx["P"])
    if force_init is not None:
        init, cold = force_init, np.zeros(len(idx), dtype=np.uint8)
    p, j, info = dev.ik_solve(torch.from_numpy(kps).to(d), torch.from_numpy(ctx["P"]).to(d),
                              torch.from_numpy(mem).to(d), torch.from_numpy(init).to(d),
                              torch.from_numpy(cold).to(d), nfev_cold, nfev_warm)
    torch.cuda.synchronize()
    return p.cpu().numpy(), j.cpu().numpy(), info.cpu().numpy()


def _well_observed(g, i, min_views=2, min_score=0.1):
    """skeleton joints that >= min_views views actually see (score > min_score); the others carry
    no residual weight, so their 3-D position is not determined by the cost at all."""
    v = int(g["n_views"][i])
    sc = np.array([o.add_mid_spine(p) for p in g["poses"][i, :v]])[:, o.IK_OBS_IDX, 2]
    return o.IK_SKEL_IDX[(sc > min_score).sum(axis=0) >= min_views]


def _obs(g, i):
    v = int(g["n_views"][i])
    obs = np.array([o.add_mid_spine(p) for p in g["poses"][i, :v]])[:, o.IK_OBS_IDX, :]
    return obs, np.asarray(g["projs"][i, :v])


def test_cost_and_fk_at_x0(ctx):
    g = ctx["g"]
    idx = ctx["idx"][~g["cold"][ctx["idx"]]][:24]
    p, j, info = _run(ctx, idx, 1, 1)
    bd, _ = o.skeleton_constants()
    for b, i in enumerate(idx):
        obs, projs = _obs(g, i)
        x0 = g["s2_x0"][i].copy()
        x0[:57] = g["s1_x0"][i]
        f = o.ik_residual(x0[:3], x0[3:57], x0[57:], obs, projs, bd)
        c = 0.5 * f.dot(f)
        assert abs(info[b, 0] - c) <= 1e-12 * c and abs(info[b, 3] - c) <= 1e-12 * c
        assert info[b, 1] == 1 and info[b, 4] == 1
        assert np.array_equal(p[b], x0)
        pos, _ = o.forward_kinematics(x0[:3], x0[3:57], x0[57:], bd)
        assert np.abs(j[b] - pos).max() < 1e-13


def test_cold_start_root_is_midpoint_of_post_optimised_hips(ctx):
    """Every cold case, 2-view clusters included (no exemption).  The reference's post-optimisation (least_squares(max_nfev=2) on an
    unsigned residual, mv_math_util.py:189-210) is one trial step that is almost always rejected, so the root is the DLT root and must
    match to 1e-5.  Where the reference ACCEPTS the step -- a rank-deficient minimum-norm Gauss-Newton step stretched to |p| = |x0|,
    metres long -- SciPy's own result moves by ~1e-5 of that length under 1-ulp changes of the start point (its LAPACK noise triplets
    take a small, rounding-dependent share of the step: tests/test_ik_sensitivity.py), so the bar there is 1e-3 of the move."""
    from helpers import twin_postopt_root
    g = ctx["g"]
    idx = ctx["idx"][g["cold"][ctx["idx"]]]
    p, j, info = _run(ctx, idx, 1, 1)
    pf, _, _ = _run_fd(ctx, idx, 1, 1)
    _, side = o.skeleton_constants()
    n_moved = 0
    for b, i in enumerate(idx):
        v = int(g["n_views"][i])
        poses18 = [o.add_mid_spine(q) for q in g["poses"][i, :v]]
        p3d = o.triangulate_groups(g["projs"][i, :v], poses18, 0.01, True)
        p3d0 = o.triangulate_groups(g["projs"][i, :v], poses18, 0.01, False)
        root = 0.5 * (p3d[11, :3] + p3d[12, :3])
        move = np.abs(root - 0.5 * (p3d0[11, :3] + p3d0[12, :3])).max()
        err, err_fd = np.abs(p[b, :3] - root).max(), np.abs(pf[b, :3] - root).max()
        assert np.abs(root - g["s1_x0"][i][:3]).max() < 1e-12   # the oracle's root is the reference's
        assert not p[b, 3:57].any() and np.array_equal(p[b, 57:], side)
        tol = 1e-5
        if move > 0:
            n_moved += 1
            tol = max(1e-5, 1e-3 * move)
            host = np.abs(twin_postopt_root(g["projs"][i, :v], poses18) - root).max()
            print(f"cold case {i} ({v} views): the reference accepted its post-optimisation step (root moved {move:.2e} m); from the "
                  f"reference: production {err:.2e}, TRF-faithful on the device {err_fd:.2e}, TRF-faithful on the host {host:.2e}")
        assert err < tol, (i, v, err, move)
        assert err_fd < tol, (i, v, err_fd, move)
    print("cold starts checked:", len(idx), "| reference accepted the post-optimisation step:", n_moved)
    assert n_moved >= 1


def _cpu_device_algorithm(g, i, x0_57, blens, nfev):
    obs, projs = _obs(g, i)
    bd, _ = o.skeleton_constants()
    f1 = lambda x: o.ik_residual(x[:3], x[3:].reshape(-1, 3), blens, obs, projs, bd)
    j1 = lambda x, f: t.ik_jacobian(x[:3], x[3:], blens, obs, projs, False)
    r1 = t.trf(f1, j1, x0_57, nfev, solver="ne")
    f2 = lambda x: o.ik_residual(x[:3], x[3:57].reshape(-1, 3), x[57:], obs, projs, bd)
    j2 = lambda x, f: t.ik_jacobian(x[:3], x[3:57], x[57:], obs, projs, True)
    r2 = t.trf(f2, j2, np.concatenate([r1["x"], blens]), nfev, solver="ne")
    pos, _ = o.forward_kinematics(r2["x"][:3], r2["x"][3:57], r2["x"][57:], bd)
    return r1, r2, pos


def test_trust_region_steps_match_cpu_restatement(ctx):
    """A trust-region step with an ACTIVE constraint (|GN step| > Delta_0 = |x0|, so alpha is
    far above the rounding noise of the null space and the step is well defined): the HIP solver and the
    CPU restatement of the same algorithm (oracle/trf_np.py) must agree to rounding."""
    from multiview_motion_capture_amd import synth
    dev, d = ctx["dev"], ctx["d"]
    rng = np.random.default_rng(5)
    K, Rt, P = synth.make_cameras(5, rng)
    bd, side = o.skeleton_constants()
    B = 12
    kps = np.zeros((B, 5, 1, 17, 3))
    init = np.zeros((B, 68))
    truth = []
    for b in range(B):
        root = rng.normal(0, 0.05, 3) + np.array([0, 0, 0.05])
        ang = rng.normal(0, 0.25, (18, 3))
        pos, _ = o.forward_kinematics(root, ang, side, bd)
        # COCO-17 observation of the pose: hips/knees/.../ears from the skeleton, eyes = nose
        coco = np.zeros((17, 3))
        for sk_j, co_j in zip(o.REPROJ_SKEL_IDX, o.REPROJ_COCO_IDX):
            coco[co_j] = pos[sk_j]
        coco[1] = coco[2] = coco[0]
        for c in range(5):
            h = P[c] @ np.concatenate([coco, np.ones((17, 1))], axis=1).T
            kps[b, c, 0, :, :2] = (h[:2] / h[2]).T + rng.normal(0, 1.0, (17, 2))
            kps[b, c, 0, :, 2] = rng.uniform(0.5, 1.0, 17)
        x0 = np.concatenate([root + rng.normal(0, 0.02, 3), (ang * 0.2).ravel(), side])  # far from the answer, small norm
        init[b] = x0
        truth.append((root, ang))
    mem = np.arange(B * 5, dtype=np.int32).reshape(B, 5)
    for nfev in (2,):  # one step: later steps run with an inactive constraint (noise-dependent, see below)
        p, j, info = dev.ik_solve(torch.from_numpy(kps).to(d), torch.from_numpy(P).to(d), torch.from_numpy(mem).to(d),
                                  torch.from_numpy(init).to(d), torch.zeros(B, dtype=torch.uint8, device=d), 50, nfev)
        p, j, info = p.cpu().numpy(), j.cpu().numpy(), info.cpu().numpy()
        worst_c = worst_j = 0.0
        n_acc = 0
        for b in range(B):
            obs = np.array([o.add_mid_spine(kps[b, c, 0]) for c in range(5)])[:, o.IK_OBS_IDX, :]
            f1 = lambda x: o.ik_residual(x[:3], x[3:].reshape(-1, 3), side, obs, P, bd)
            j1 = lambda x, f: t.ik_jacobian(x[:3], x[3:], side, obs, P, False)
            tr = []
            r1 = t.trf(f1, j1, init[b, :57], nfev, solver="ne", trace=tr)
            assert tr[0]["alpha"] > 1e-3, "test premise: the trust-region constraint is active"
            n_acc += int(tr[0]["accepted"])
            f2 = lambda x: o.ik_residual(x[:3], x[3:57].reshape(-1, 3), x[57:], obs, P, bd)
            j2 = lambda x, f: t.ik_jacobian(x[:3], x[3:57], x[57:], obs, P, True)
            r2 = t.trf(f2, j2, np.concatenate([r1["x"], side]), nfev, solver="ne")
            pos, _ = o.forward_kinematics(r2["x"][:3], r2["x"][3:57], r2["x"][57:], bd)
            # stage 1 is the well-defined one (stage 2 starts with Delta_0 >= |side lengths| ~ 1 m, an
            # inactive constraint, and then depends on null-space rounding noise -- DESIGN.md "IK parity")
            worst_c = max(worst_c, abs(info[b, 0] - r1["cost"]) / r1["cost"])
            worst_j = max(worst_j, abs(info[b, 3] - r2["cost"]) / r2["cost"])
            assert info[b, 1] == r1["nfev"]
        print(f"trust-region steps (max_nfev={nfev}): accepted first steps {n_acc}/{B}, stage-1 worst rel cost diff "
              f"{worst_c:.2e} (stage-2, informational: {worst_j:.2e})")
        assert n_acc >= B // 2
        assert worst_c < 1e-7


def test_converged_cold_starts_match_reference(ctx):
    """Well-posed cold starts (>= 3 views, both stages terminated by ftol/xtol in the reference)."""
    g = ctx["g"]
    sel = [i for i in ctx["idx"] if g["cold"][i] and g["n_views"][i] >= 3 and g["s1_status"][i] > 0
           and g["s2_status"][i] > 0]
    assert len(sel) >= 2
    p, j, info = _run(ctx, np.array(sel), 50, 5)
    for b, i in enumerate(sel):
        rel_cost = abs(info[b, 3] - g["s2_cost"][i]) / g["s2_cost"][i]
        obs_joints = _well_observed(g, i, min_views=3)
        dj = np.abs(j[b][obs_joints] - g["joints"][i][obs_joints]).max()
        scale = np.abs(g["joints"][i]).max()
        print("cold case", i, "views", g["n_views"][i], "rel cost", rel_cost, "joint diff", dj, "status", info[b, [2, 5]])
        assert rel_cost < 1e-4
        assert dj / scale < 1e-4  # north_star tolerance on the 3-D joints the views observe


def _reference_band(g, idx):
    """How far float-equivalent implementations of the REFERENCE'S OWN algorithm land from the reference on the truncated warm
    solves (5 + 5 evaluations, status 0): (a) SciPy least_squares on a residual whose projection product is one einsum instead of
    per-view matmuls; (b) oracle/trf_np.py, the NumPy restatement of SciPy's TRF (2-point Jacobian, SVD step).  Same mathematics,
    different rounding.  -> {name: joint distances (m) on the joints >= 2 views see}."""
    from scipy.optimize import least_squares
    bd, _ = o.skeleton_constants()

    def res_einsum(root, euler, blens, obs, projs):
        pos, _ = o.forward_kinematics(root, euler, blens, bd)
        X = pos[o.IK_SKEL_IDX]
        h = np.einsum('vik,jk->vji', projs, np.concatenate([X, np.ones((len(X), 1))], axis=1))
        return (((h[..., :2] / (1e-5 + h[..., 2:3])) - obs[..., :2]) * obs[..., 2:3]).ravel()

    out = {"scipy_einsum": [], "numpy_trf": []}
    for i in idx:
        obs, projs = _obs(g, i)
        bl = g["init_blens"][i]
        wo = _well_observed(g, i)
        r1 = least_squares(lambda x: res_einsum(x[:3], x[3:].reshape(-1, 3), bl, obs, projs), g["s1_x0"][i], max_nfev=5)
        r2 = least_squares(lambda x: res_einsum(x[:3], x[3:57].reshape(-1, 3), x[57:], obs, projs),
                           np.concatenate([r1.x, bl]), max_nfev=5)
        pos, _ = o.forward_kinematics(r2.x[:3], r2.x[3:57], r2.x[57:], bd)
        out["scipy_einsum"].append(np.abs(pos[wo] - g["joints"][i][wo]).max())
        f1 = lambda x: o.ik_residual(x[:3], x[3:].reshape(-1, 3), bl, obs, projs, bd)
        f2 = lambda x: o.ik_residual(x[:3], x[3:57].reshape(-1, 3), x[57:], obs, projs, bd)
        q1 = t.trf(f1, lambda x, f: t.fd_jacobian(f1, x, f), g["s1_x0"][i], 5)
        q2 = t.trf(f2, lambda x, f: t.fd_jacobian(f2, x, f), np.concatenate([q1["x"], bl]), 5)
        pos, _ = o.forward_kinematics(q2["x"][:3], q2["x"][3:57], q2["x"][57:], bd)
        out["numpy_trf"].append(np.abs(pos[wo] - g["joints"][i][wo]).max())
    return {k: np.array(v) for k, v in out.items()}


@pytest.fixture(scope="module")
def band(ctx):
    g = ctx["g"]
    idx = ctx["idx"][~g["cold"][ctx["idx"]]]
    b = _reference_band(g, idx)
    for k, v in b.items():
        print("reference band [%s]: joints median %.2e p90 %.2e max %.2e" % (k, np.median(v), np.quantile(v, 0.9), v.max()))
    return b


def _warm_distances(ctx, runner):
    g = ctx["g"]
    idx = ctx["idx"][~g["cold"][ctx["idx"]]]
    p, j, info = runner(ctx, idx, 50, 5)
    dj = np.array([np.abs(j[b][_well_observed(g, i)] - g["joints"][i][_well_observed(g, i)]).max() for b, i in enumerate(idx)])
    rc = np.array([(info[b, 3] - g["s2_cost"][i]) / g["s2_cost"][i] for b, i in enumerate(idx)])
    return dj, rc, info


def test_truncated_warm_solves_within_reference_sensitivity_band(ctx, band):
    """max_nfev = 5 + 5 (status 0 in the reference: cut off, not converged).  The reference's answer here is a function of rounding:
    implementations of ITS OWN algorithm that differ only in floating-point summation order land 2e-3 .. 4e-3 m (median) away from
    it (`band`, computed here on the same 45 cases).  The production solver (analytic Jacobian, Krylov step) must be inside that
    band -- no farther from the reference than the reference's float-equivalent twins are -- and not systematically worse in cost."""
    dj, rc, info = _warm_distances(ctx, _run)
    print("device (production): joint diff median %.2e p90 %.2e max %.2e ; rel cost median|.| %.2e mean %.2e max %.2e min %.2e" %
          (np.median(dj), np.quantile(dj, 0.9), dj.max(), np.median(np.abs(rc)), rc.mean(), rc.max(), rc.min()))
    assert (info[:, 1] <= 5).all() and (info[:, 4] <= 5).all()
    med = max(np.median(v) for v in band.values())
    p90 = max(np.quantile(v, 0.9) for v in band.values())
    assert np.median(dj) <= 1.25 * med, (np.median(dj), med)
    assert np.quantile(dj, 0.9) <= 1.5 * p90, (np.quantile(dj, 0.9), p90)   # a tail statistic of 45 samples
    assert np.median(np.abs(rc)) < 1e-2
    # not systematically worse than the reference (single truncated solves can land far apart either way)
    assert np.median(rc) < 5e-3 and (rc > 0.1).mean() < 0.1


def _run_fd(ctx, idx, nfev_cold, nfev_warm):
    dev, d = ctx["dev"], ctx["d"]
    kps, mem, init, cold = _pack(ctx["g"], idx, ctx["P"])
    p, j, info = dev.ik_solve_fd(torch.from_numpy(kps).to(d), torch.from_numpy(ctx["P"]).to(d), torch.from_numpy(mem).to(d),
                                 torch.from_numpy(init).to(d), torch.from_numpy(cold).to(d), nfev_cold, nfev_warm)
    torch.cuda.synchronize()
    return p.cpu().numpy(), j.cpu().numpy(), info.cpu().numpy()


def test_trf_faithful_mode_lands_in_the_same_band(ctx, band):
    """The device's TRF-faithful solver (mvmc_debug_ik_solve_fd: 2-point finite-difference Jacobian, SVD trust-region step -- the
    reference's method, csrc/mvmc_trf_faithful.h) on the same cases: it is one more float-equivalent twin of the reference, lands in
    the same band, and so shows that the production solver's distance is the rounding chaos of the truncated solve and not its
    analytic Jacobian / Krylov step."""
    dj, rc, info = _warm_distances(ctx, _run_fd)
    dj_prod, _, _ = _warm_distances(ctx, _run)
    print("device (TRF-faithful): joint diff median %.2e p90 %.2e max %.2e ; rel cost median|.| %.2e mean %.2e" %
          (np.median(dj), np.quantile(dj, 0.9), dj.max(), np.median(np.abs(rc)), rc.mean()))
    print("device production vs TRF-faithful medians: %.2e vs %.2e" % (np.median(dj_prod), np.median(dj)))
    assert (info[:, 1] <= 5).all() and (info[:, 4] <= 5).all() and (info[:, 1] >= 2).all()
    med = max(np.median(v) for v in band.values())
    p90 = max(np.quantile(v, 0.9) for v in band.values())
    assert np.median(dj) <= 1.5 * med and np.quantile(dj, 0.9) <= 1.5 * p90
    # the faithful twin is no closer to the reference than twice the production solver: the gap is not the Jacobian's
    assert np.median(dj_prod) <= 2.0 * np.median(dj) + 1e-3


def test_trf_faithful_mode_converged_cold_starts(ctx):
    """Converged cold starts through the TRF-faithful solver (incl. its own post-optimised DLT root): the reference's minimum."""
    g = ctx["g"]
    sel = [i for i in ctx["idx"] if g["cold"][i] and g["n_views"][i] >= 3 and g["s1_status"][i] > 0 and g["s2_status"][i] > 0]
    p, j, info = _run_fd(ctx, np.array(sel), 50, 5)
    for b, i in enumerate(sel):
        rel_cost = abs(info[b, 3] - g["s2_cost"][i]) / g["s2_cost"][i]
        oj = _well_observed(g, i, min_views=3)
        dj = np.abs(j[b][oj] - g["joints"][i][oj]).max()
        print("TRF-faithful cold case", i, "rel cost", rel_cost, "joint diff", dj, "status", info[b, [2, 5]], "nfev", info[b, [1, 4]])
        assert rel_cost < 1e-4 and dj / np.abs(g["joints"][i]).max() < 1e-4
