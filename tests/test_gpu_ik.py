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
    g = ctx["g"]
    idx = ctx["idx"][g["cold"][ctx["idx"]]][:8]
    p, j, info = _run(ctx, idx, 1, 1)
    _, side = o.skeleton_constants()
    for b, i in enumerate(idx):
        v = int(g["n_views"][i])
        poses18 = [o.add_mid_spine(q) for q in g["poses"][i, :v]]
        p3d = o.triangulate_groups(g["projs"][i, :v], poses18, 0.01, True)
        root = 0.5 * (p3d[11, :3] + p3d[12, :3])
        if v >= 3:  # 2-view post-optimise is rank deficient (noise-driven step), see DESIGN.md
            assert np.abs(p[b, :3] - root).max() < 1e-5, (i, v, np.abs(p[b, :3] - root).max())
        assert np.abs(p[b, :3] - g["s1_x0"][i][:3]).max() < 1e-5 or v < 3
        assert not p[b, 3:57].any() and np.array_equal(p[b, 57:], side)


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


def test_truncated_warm_solves_within_reference_sensitivity_band(ctx):
    """max_nfev = 5 + 5 (status 0 in the reference): compared inside the band in which the reference
    itself moves under a float-equivalent reformulation (tests/test_ik_sensitivity.py: up to ~1e-2 m,
    ~3 % in cost).  Also requires that the device solve is not systematically worse."""
    g = ctx["g"]
    idx = ctx["idx"][~g["cold"][ctx["idx"]]]
    p, j, info = _run(ctx, idx, 50, 5)
    dj = np.array([np.abs(j[b][_well_observed(g, i)] - g["joints"][i][_well_observed(g, i)]).max()
                   for b, i in enumerate(idx)])
    rc = np.array([(info[b, 3] - g["s2_cost"][i]) / g["s2_cost"][i] for b, i in enumerate(idx)])
    print("warm: joint diff median %.2e p90 %.2e max %.2e ; rel cost median|.| %.2e mean %.2e max %.2e min %.2e" %
          (np.median(dj), np.quantile(dj, 0.9), dj.max(), np.median(np.abs(rc)), rc.mean(), rc.max(), rc.min()))
    assert (info[:, 1] <= 5).all() and (info[:, 4] <= 5).all()
    assert np.median(dj) < 1e-2 and np.quantile(dj, 0.9) < 3e-2
    assert np.median(np.abs(rc)) < 1e-2
    # not systematically worse than the reference (single truncated solves can land far apart either way)
    assert np.median(rc) < 5e-3 and (rc > 0.1).mean() < 0.1
