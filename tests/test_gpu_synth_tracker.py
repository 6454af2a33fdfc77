"""GPU: the persistent chain kernel (mvmc_chain_run, the benchmark's default path) against the REFERENCE tracker on the benchmark's own
workloads: tests/golden/synth_c4_tracker.npz and synth_c5_tracker.npz hold MvTracker.update_4d (motion_capture.py:873-963) run by the
reference itself over 64-frame subsets of synthetic config 4 (seed 20260103, C5 P4: the SMALL layout) and config 5 (seed 20260104,
C8 P8: the BIG layout, als5), chains of 16 (oracle/gen_golden_ikconv.py, oracle/gen_golden_c5.py) -- SURVEY.md section 8c."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fixture", ["synth_c4_tracker.npz", "synth_c5_tracker.npz"])
def test_chain_kernel_reproduces_the_reference_tracker_on_the_synthetic_workload(fixture):
    from multiview_motion_capture_amd import synth
    from multiview_motion_capture_amd.pipeline import HotPath
    from multiview_motion_capture_amd.tracker import check_chain_flags, run_chains_fused
    g = load_golden(fixture)
    F, L, C, P = int(g["n_frames"]), int(g["chain_len"]), int(g["n_views"]), int(g["n_people"])
    data = synth.generate(F, C, P, int(g["seed"]), chain_len=L)
    assert float(np.abs(data["kps25"].astype(np.float64)).sum()) == float(g["kps25_checksum"])   # same inputs as the reference saw
    d = torch.device("cuda:0")
    hp = HotPath(data["K"], data["Rt"], device=d)
    out = run_chains_fused(hp, torch.from_numpy(data["kps25"]).to(d), torch.from_numpy(data["counts"]).to(d), L, want_info=True)
    torch.cuda.synchronize()
    check_chain_flags(out)
    n_t, meta, joints = out["n_tracks"].cpu().numpy(), out["meta"].cpu().numpy(), out["joints"].cpu().numpy()
    # ---- association + tracker state machine: exact ----
    assert np.array_equal(n_t, g["n_tracks"])
    for f in range(F):
        k = int(g["n_tracks"][f])
        assert np.array_equal(meta[f, :k], g["meta"][f, :k]), (f, meta[f, :k], g["meta"][f, :k])
    n_dead = out["n_dead"].cpu().numpy()
    assert np.array_equal(n_dead, g["n_dead"][L - 1::L])
    info = out["ik_info"].cpu().numpy().reshape(F, -1, 8)
    solved = ~np.isnan(info[:, :, 1])
    assert np.array_equal(solved.sum(axis=1), g["n_solves"])
    # ---- 3-D output: cold chain heads are converged solves (tight); warm frames are 5 + 5-evaluation truncated solves ----
    dj = np.full((F, P), np.nan)
    for f in range(F):
        for s in range(int(g["n_tracks"][f])):
            dj[f, s] = np.abs(joints[f, s] - g["joints"][f, s]).max()
    head = dj[0::L].ravel()
    warm = np.concatenate([dj[b * L + 1:(b + 1) * L].ravel() for b in range(F // L)])
    print("chain heads (cold, 50 + 50): joint diff median %.2e max %.2e | warm frames (5 + 5): median %.2e p90 %.2e max %.2e" %
          (np.nanmedian(head), np.nanmax(head), np.nanmedian(warm), np.nanquantile(warm, 0.9), np.nanmax(warm)))
    gt = data["gt_joints"]
    e_dev, e_ref = [], []
    for f in range(F):
        for s in range(int(g["n_tracks"][f])):
            pj = np.linalg.norm(gt[f] - g["joints"][f, s][None], axis=-1).mean(axis=-1)
            e_ref.append(pj.min())
            e_dev.append(np.linalg.norm(gt[f, int(pj.argmin())] - joints[f, s], axis=-1).mean())
    print("mean joint error vs ground truth: device %.4f m, reference %.4f m" % (np.mean(e_dev), np.mean(e_ref)))
    assert np.nanmax(head) < 1e-5          # converged solves (observed: 9.5e-7 on config 4)
    assert np.nanmedian(warm) < 5e-3 and np.nanquantile(warm, 0.9) < 1.6e-2     # the reference's own rounding band (tests/test_gpu_ik.py)
    assert np.mean(e_dev) < 1.1 * np.mean(e_ref) + 1e-3


@pytest.mark.parametrize("C,P,seed,n_chains", [(5, 4, 20260103, 3), (8, 8, 20260104, 1)])
def test_chain_kernel_equals_the_noise_free_oracle_tracker_on_every_frame(C, P, seed, n_chains):
    """The same workloads against the DETERMINISTIC oracle (tracker_np.OracleTracker driving trf_np.pose_solver_solve_clean: the
    reference's tracker with its two least_squares calls freed of LAPACK's noise, tests/test_gpu_tracker.py): the warm frames, which
    the comparison with the reference's recorded results above can only hold to a band, must agree like the chain heads -- tracker
    tables on every frame, joints to 1e-8."""
    import oracle_np as o
    import tracker_np as tk
    import trf_np as t
    from multiview_motion_capture_amd import synth
    from multiview_motion_capture_amd.pipeline import HotPath
    from multiview_motion_capture_amd.tracker import check_chain_flags, run_chains_fused
    L = 16
    F = n_chains * L
    data = synth.generate(F, C, P, seed, chain_len=L)
    d = torch.device("cuda:0")
    hp = HotPath(data["K"], data["Rt"], device=d)
    out = run_chains_fused(hp, torch.from_numpy(data["kps25"]).to(d), torch.from_numpy(data["counts"]).to(d), L)
    torch.cuda.synchronize()
    check_chain_flags(out)
    n_t, meta, joints = out["n_tracks"].cpu().numpy(), out["meta"].cpu().numpy(), out["joints"].cpu().numpy()
    kps25 = data["kps25"].astype(np.float64)
    dd = []
    for b in range(n_chains):
        orc = tk.OracleTracker(data["K"], data["Rt"], data["P"], solver=lambda poses, projs, init: t.pose_solver_solve_clean(poses, projs, init))
        for tt in range(L):
            f = b * L + tt
            views = []
            for c in range(C):
                poses = [o.openpose25_to_coco17(kps25[f, c, p]) for p in range(int(data["counts"][f, c]))]
                views.append([q for q in poses if o.pose_is_good(q)])
            orc.update(tt, views)
            exp = np.array([[x.tid, x.state, x.hits, x.length] for x in orc.tracklets], dtype=np.int32).reshape(-1, 4)
            assert n_t[f] == len(exp) and np.array_equal(meta[f, :len(exp)], exp), (f, meta[f, :n_t[f]], exp)
            dd += [float(np.abs(joints[f, s] - x.joints).max()) for s, x in enumerate(orc.tracklets)]
        assert orc.n_dead == int(out["n_dead"][b]) and orc.next_id == int(out["next_id"][b])
    dd = np.array(dd)
    print(f"\nC{C} P{P}: {n_chains} chain(s) of {L} frames against the noise-free oracle tracker: tables equal on every frame; {len(dd)} "
          f"tracklet-frames, joint difference median {np.median(dd):.1e} p90 {np.percentile(dd, 90):.1e} max {dd.max():.1e} m")
    # observed: 1.2e-13 m (config 4), 5.3e-11 m (config 5) at worst -- every observation is seen by all views here, no weak models
    assert dd.max() < 1e-8
