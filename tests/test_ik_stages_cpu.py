"""Oracle restatement of the 3-D-target IK variants (solve_pose / solve_pose_bone_lens, inverse_kinematics.py:280-336)
pinned to vectors made by running the reference itself (oracle/gen_golden_ik3d.py -> tests/golden/ik3d_cases.npz)."""
import numpy as np

import oracle_np as o
from conftest import load_golden


def test_oracle_3d_stages_reproduce_the_reference():
    g = load_golden("ik3d_cases.npz")
    bd, side = o.skeleton_constants()
    assert len(g["case"]) == 20
    for i in range(0, len(g["case"]), 3):
        x0, nfev = g["x0"][i], int(g["nfev"][i])
        r1, e1, _ = o.ik_stage1_3d(g["obs3d"][i], x0[:3], x0[3:57].reshape(18, 3), x0[57:], nfev, bd)
        assert np.abs(np.concatenate([r1, e1.ravel()]) - g["x1"][i][:57]).max() < 1e-9
        r2, e2, b2, _ = o.ik_stage2_3d(g["obs3d"][i], r1, e1, x0[57:], nfev, bd)
        assert np.abs(np.concatenate([r2, e2.ravel(), b2]) - g["x2"][i]).max() < 1e-9
        joints, _ = o.forward_kinematics(r2, e2, b2, bd)
        assert np.abs(joints - g["joints"][i]).max() < 1e-9
