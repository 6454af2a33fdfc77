"""CPU: (1) the oracle of the temporal layer (oracle/tracker_np.py: match_spatial_time + tracker state
machine + IK) reproduces the reference's Shelf log exactly; (2) evidence that the log is not
reproducible beyond ~90 frames by the reference itself once its residual is re-ordered in a
float-equivalent way (the bar used for the device tracker in tests/test_gpu_tracker.py)."""
import numpy as np

import oracle_np as o
import tracker_np as tk
from conftest import load_golden
from helpers import oracle_ingest


def _run(n_frames, residual=None):
    si, g = load_golden("shelf_inputs.npz"), load_golden("shelf_tracker.npz")
    k17, cnt = oracle_ingest(si["kps25"][:n_frames + 1], si["counts"][:n_frames + 1].astype(np.int32))
    orig = o.ik_residual
    if residual is not None:
        o.ik_residual = residual
    try:
        tr = tk.OracleTracker(si["K"], si["Rt"], si["P"])
        same, first = 0, None
        for fi in range(1, n_frames + 1):
            views = [[k17[fi, c, p] for p in range(cnt[fi, c])] for c in range(5)]
            n0 = len(tr.solves)
            tr.update(fi, views)
            exp = [tuple(int(v) for v in r) for r in g["alive_after"][fi - 1] if r[0] >= 0]
            got = [(t.tid, t.state, t.hits, t.length) for t in tr.tracklets]
            ok = got == exp and tr.n_dead == g["n_dead"][fi - 1] and len(tr.solves) - n0 == g["n_solves"][fi - 1]
            same += int(ok)
            if not ok and first is None:
                first = fi
    finally:
        o.ik_residual = orig
    return tr, same, first, g


def test_oracle_tracker_reproduces_reference_log_exactly():
    n = 60
    tr, same, first, g = _run(n)
    assert same == n and first is None
    roots = np.array([s[3][0] for s in tr.solves])
    joints = np.array([s[4] for s in tr.solves])
    assert np.array_equal(roots, g["solves"][:len(roots), 3:6])     # bit-exact IK trajectories
    assert np.array_equal(joints, g["solve_joints"][:len(joints)])
    assert [int(s[1]) for s in tr.solves] == [int(c) for c in g["solves"][:len(roots), 1]]


def _residual_einsum(root, euler, side_blens, obs, projs, bone_dirs=None):
    pos, _ = o.forward_kinematics(root, euler, side_blens, bone_dirs)
    X = pos[o.IK_SKEL_IDX]
    h = np.einsum('vik,jk->vji', projs, np.concatenate([X, np.ones((len(X), 1))], axis=1))
    uv = h[..., :2] / (1e-5 + h[..., 2:3])
    return ((uv - obs[..., :2]) * obs[..., 2:3]).ravel()


def test_reference_log_is_not_reproducible_under_float_reordering():
    n = 100
    _, same, first, _ = _run(n, _residual_einsum)
    print("reference vs float-equivalent reference: identical tracker state on", same, "of", n, "frames; first divergence", first)
    assert first is not None and 60 <= first <= n  # observed: frame 92


def test_noise_free_oracle_tracker_agrees_with_the_reference_log_where_that_is_reproducible():
    """The whole-sequence oracle of tests/test_gpu_tracker.py -- OracleTracker driving trf_np.pose_solver_solve_clean, the reference's two
    least_squares calls without LAPACK's noise in the null directions -- pinned to the reference where the reference is reproducible: the
    tracker tables (ids, states, hits, lengths, deaths, solves per frame) of the reference's own log over the first 40 Shelf frames, and its
    solved joints inside the band in which the reference's float-equivalent twins land (tests/test_ik_sensitivity.py)."""
    import trf_np as t
    n = 40
    si, g = load_golden("shelf_inputs.npz"), load_golden("shelf_tracker.npz")
    k17, cnt = oracle_ingest(si["kps25"][:n + 1], si["counts"][:n + 1].astype(np.int32))
    tr = tk.OracleTracker(si["K"], si["Rt"], si["P"], solver=lambda poses, projs, init: t.pose_solver_solve_clean(poses, projs, init))
    fx = load_golden("shelf_clean_oracle_tracker.npz")     # the same oracle over all 300 frames (oracle/gen_golden_shelf_clean.py)
    fx_worst = 0.0
    for fi in range(1, n + 1):
        views = [[k17[fi, c, p] for p in range(cnt[fi, c])] for c in range(5)]
        n0 = len(tr.solves)
        tr.update(fi, views)
        exp = [tuple(int(v) for v in r) for r in g["alive_after"][fi - 1] if r[0] >= 0]
        assert [(x.tid, x.state, x.hits, x.length) for x in tr.tracklets] == exp, fi
        assert tr.n_dead == g["n_dead"][fi - 1] and len(tr.solves) - n0 == g["n_solves"][fi - 1], fi
        # ... and the committed whole-sequence fixture is what this oracle gives here (another host's LAPACK may differ in the last bits)
        assert fx["n_tracks"][fi - 1] == len(exp) and [tuple(int(v) for v in r) for r in fx["meta"][fi - 1, :len(exp)]] == exp, fi
        for s_, x in enumerate(tr.tracklets):
            fx_worst = max(fx_worst, float(np.abs(fx["joints"][fi - 1, s_] - x.joints).max()))
    print(f"fixture shelf_clean_oracle_tracker.npz vs the oracle run here, {n} frames: worst joint difference {fx_worst:.1e} m")
    assert fx_worst < 1e-7
    joints = np.array([s[4] for s in tr.solves])
    ref = g["solve_joints"][:len(joints)]
    d = np.abs(joints - ref)[:, o.IK_SKEL_IDX].max(axis=(1, 2))
    print(f"noise-free oracle tracker vs the reference's log, {n} frames, {len(d)} solves: observed joints median {np.median(d):.2e} m, max {d.max():.2e} m")
    assert np.median(d) < 1e-2 and d.max() < 0.1
