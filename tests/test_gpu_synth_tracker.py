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
