"""Golden fixture tests/golden/shelf_clean_oracle_tracker.npz: the noise-free oracle tracker on the 300 Shelf frames.

TEST INFRASTRUCTURE.  The oracle here is tracker_np.OracleTracker (match_spatial_time + MvTracker.update_4d restated; bit-exact against the
reference's own tracker log with SciPy's solver, tests/test_tracker_oracle_cpu.py) driving trf_np.pose_solver_solve_clean (PoseSolver.solve,
inverse_kinematics.py:380-433, with its two least_squares calls as trf(solver="ne_clean"): SciPy's TRF on the normal equations with the
null cluster dropped -- pinned against SciPy's own steps in tests/test_trf_traces_cpu.py).  It needs nothing of /root/reference: inputs are
tests/golden/shelf_inputs.npz (made by gen_golden.py from the reference's data/shelf files).  ~100 s on one core.

    python oracle/gen_golden_shelf_clean.py            # writes the fixture
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle_np as o      # noqa: E402
import tracker_np as tk    # noqa: E402
import trf_np as t         # noqa: E402

T_MAX = 8


def run(n_frames=300, first=1):
    si = np.load(os.path.join(HERE, "..", "tests", "golden", "shelf_inputs.npz"))
    orc = tk.OracleTracker(si["K"], si["Rt"], si["P"], solver=lambda poses, projs, init: t.pose_solver_solve_clean(poses, projs, init))
    n = np.zeros(n_frames, np.int32)
    meta = np.full((n_frames, T_MAX, 4), -1, np.int32)
    joints = np.full((n_frames, T_MAX, 18, 3), np.nan)
    params = np.full((n_frames, T_MAX, 68), np.nan)
    for k in range(n_frames):
        fi = first + k
        views = []
        for c in range(5):
            poses = [o.openpose25_to_coco17(si["kps25"][fi, c, p]) for p in range(int(si["counts"][fi, c]))]
            views.append([p for p in poses if o.pose_is_good(p)])
        orc.update(fi, views)
        n[k] = len(orc.tracklets)
        for s, tr in enumerate(orc.tracklets):
            meta[k, s] = (tr.tid, tr.state, tr.hits, tr.length)
            joints[k, s] = tr.joints
            params[k, s] = np.concatenate([tr.param[0], tr.param[1].ravel(), tr.param[2]])
    return dict(n_tracks=n, meta=meta, joints=joints, params=params, next_id=np.int32(orc.next_id), n_dead=np.int32(orc.n_dead),
                first_frame=np.int32(first))


if __name__ == "__main__":
    out = os.path.join(HERE, "..", "tests", "golden", "shelf_clean_oracle_tracker.npz")
    np.savez_compressed(out, **run())
    print("wrote", out, os.path.getsize(out), "bytes")
