"""Golden vectors for the 3-D-target IK variants, made by running the reference's own
solve_pose / solve_pose_bone_lens (inverse_kinematics.py:280-336) in this container through oracle/ref_shim.py, on
clusters taken from tests/golden/ik_cases.npz.  Test infrastructure only.

    python oracle/gen_golden_ik3d.py            ->  tests/golden/ik3d_cases.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    m = ref_shim.load_modules()
    g = np.load(os.path.join(OUT, "ik_cases.npz"))
    skel = m.ik.load_skeleton()
    sel = [i for i in range(len(g["frame"])) if g["n_views"][i] >= 3][:10]
    rows = []
    for i in sel:
        v = int(g["n_views"][i])
        poses = [g["poses"][i, s] for s in range(v)]
        projs = [g["projs"][i, s] for s in range(v)]
        solver = m.ik.PoseSolver(skel, None, poses, projs, m.pose_def.KpsFormat.COCO)
        obs3d = m.mu.triangulate_point_groups_from_multiple_views_linear(solver.cam_projs, solver.cam_poses_2d, 0.01, True)
        root = 0.5 * (obs3d[11, :3] + obs3d[12, :3])
        init = m.ik.PoseShapeParam(root, np.zeros((18, 3)), skel.ref_side_bone_lens.copy())
        for nfev in (50, 5):
            p1 = m.ik.solve_pose(skel, obs3d, solver.obs_kps_idxs, solver.skel_kps_idxs, init, nfev)
            p2 = m.ik.solve_pose_bone_lens(skel, obs3d, solver.obs_kps_idxs, solver.skel_kps_idxs, p1, nfev)
            j2, _ = m.ik.foward_kinematics(skel, p2)
            rows.append(dict(case=i, nfev=nfev, obs3d=obs3d, x0=np.concatenate([root, np.zeros(54), init.bone_lens]),
                             x1=np.concatenate([p1.root, p1.euler_angles.ravel(), p1.bone_lens]),
                             x2=np.concatenate([p2.root, p2.euler_angles.ravel(), p2.bone_lens]), joints=j2))
        assert list(solver.obs_kps_idxs) == [11, 13, 15, 12, 14, 16, 17, 5, 7, 9, 6, 8, 10, 0, 3, 4]
        assert list(solver.skel_kps_idxs) == [1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15, 16, 17]
    np.savez_compressed(os.path.join(OUT, "ik3d_cases.npz"), **{k: np.array([r[k] for r in rows]) for k in rows[0]})
    print("wrote", len(rows), "rows")


if __name__ == "__main__":
    main()
