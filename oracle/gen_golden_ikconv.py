"""Golden vectors of CONVERGED reference IK solves and of the reference tracker on the benchmark's own synthetic workload.

TEST INFRASTRUCTURE (build container only).  Runs the reference's own code from /root/reference/src through
``oracle/ref_shim.py``; only inputs and outputs are written (.npz), no reference source travels.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_ikconv.py [--shelf-frames 150] [--max-nfev 400] [--procs 8]

Runs with single-threaded BLAS (set below, before NumPy loads): eight worker processes with eight OpenBLAS threads each spend their
time spinning (45 min instead of 30 s), and LAPACK's rounding depends on the thread count -- the CPU tests that compare the oracle
bit for bit with these fixtures limit BLAS to one thread the same way (tests/test_synth_pins_cpu.py).

Fixtures written under tests/golden:
  ik_converged.npz      the reference's ``solve_pose_reproj`` followed by ``solve_pose_bone_lens_reproj``
                        (inverse_kinematics.py:202-277) run with ``n_max_iter`` large enough for both least_squares calls
                        to terminate by ftol / xtol / gtol (status > 0), on
                          * Shelf clusters of 2-5 views taken from the reference tracker's own run (cold init exactly as
                            PoseSolver.solve builds it, :389-397, and the tracker's warm init), and
                          * clusters of a 64-frame subset of synthetic config 4 (seed 20260103, chains of 16).
  synth_c4_tracker.npz  MvTracker.update_4d (motion_capture.py:873-963) run by the reference itself over that 64-frame subset,
                        one fresh tracker per chain of 16 frames (the benchmark's protocol): the tracklet
                        table after every frame (ids, states, hits, lengths, parameters, joints), the solves per frame.
"""
import argparse
import multiprocessing as mp
import os
import sys
import time

os.environ["OPENBLAS_NUM_THREADS"] = "1"
os.environ["OMP_NUM_THREADS"] = "1"

import numpy as np  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_shim  # noqa: E402
import gen_golden as gg  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SYNTH = dict(n_frames=64, n_views=5, n_people=4, seed=20260103, chain_len=16)
V_PAD = 6

_M = None   # reference modules (set before the pool forks)


def _solve_converged(case):
    """One reference solve with a large evaluation budget.  case = (poses [V x (17,3)], projs [V x (3,4)], init or None, max_nfev)."""
    m = _M
    poses, projs, init, max_nfev = case
    skel = m.ik.load_skeleton()
    PoseSolver = sys.modules["inverse_kinematics_pino"].PoseSolver
    ps = PoseSolver(skel, None, [np.array(p) for p in poses], [np.array(p) for p in projs], obs_kps_format=m.pose_def.KpsFormat.COCO)
    if init is None:
        # PoseSolver.solve's cold start, inverse_kinematics.py:389-397
        p3d = m.mu.triangulate_point_groups_from_multiple_views_linear(ps.cam_projs, ps.cam_poses_2d, 0.01, True)
        KT = m.pose_def.KpsType
        root = 0.5 * (p3d[ps.obs_kps_idx_map[KT.L_Hip], :3] + p3d[ps.obs_kps_idx_map[KT.R_Hip], :3])
        init_param = m.ik.PoseShapeParam(root, np.zeros((18, 3)), skel.ref_side_bone_lens.copy())
    else:
        init_param = m.ik.PoseShapeParam(np.array(init[0]), np.array(init[1]), np.array(init[2]))
    t0 = time.time()
    with gg.LsqRecorder(m.ik) as rec:
        p1 = m.ik.solve_pose_reproj(ps.skel, np.array(ps.cam_poses_2d), ps.obs_kps_idxs, ps.cam_projs, ps.skel_kps_idxs, init_param, max_nfev)
        p2 = m.ik.solve_pose_bone_lens_reproj(ps.skel, np.array(ps.cam_poses_2d), ps.obs_kps_idxs, ps.cam_projs, ps.skel_kps_idxs, p1, max_nfev)
    joints, _ = m.ik.foward_kinematics(skel, p2)
    (x0a, ra, _), (x0b, rb, _) = rec.results
    return dict(init=(np.array(init_param.root), np.array(init_param.euler_angles), np.array(init_param.bone_lens)),
                s1=(ra.x.copy(), ra.cost, ra.nfev, ra.status), s2=(rb.x.copy(), rb.cost, rb.nfev, rb.status),
                joints=np.array(joints), secs=time.time() - t0)


def synth_frames(m, data, calibs, f):
    """FrameData list of synthetic frame f, built the way parse_openpose_kps builds it (motion_capture.py:974-984)."""
    frames = []
    C = data["kps25"].shape[1]
    for c in range(C):
        poses = {}
        for p in range(int(data["counts"][f, c])):
            kps = np.array(data["kps25"][f, c, p], dtype=np.float64)
            coco = m.pose_def.conversion_openpose_25_to_coco(kps)
            poses[p] = m.pose_def.Pose(m.pose_def.KpsFormat.COCO, keypoints=coco[:, :2],
                                       keypoints_score=coco[:, -1][:, np.newaxis], box=None)
        fd = m.mc.FrameData(f, poses, calibs[c], view_id=c + 1)
        frames.append(m.mc.filter_bad_pose(fd, 0.01, 4, 5))
    return frames


def run_synth_tracker(m, SYNTH=SYNTH, T=8):
    """The reference tracker over the synthetic subset, one tracker per chain; -> (fixture dict, IK cases).  T = rows of the tables."""
    from multiview_motion_capture_amd import synth
    data = synth.generate(SYNTH["n_frames"], SYNTH["n_views"], SYNTH["n_people"], SYNTH["seed"], chain_len=SYNTH["chain_len"])
    C = SYNTH["n_views"]
    calibs = []
    for c in range(C):
        K, Rt = data["K"][c], data["Rt"][c]
        calibs.append(m.common.Calib(K=K, Rt=Rt, P=K @ Rt, Kr_inv=Rt[:, :3].T @ np.linalg.inv(K), img_wh_size=(1032, 776)))
    skel = m.ik.load_skeleton()
    PoseSolver = sys.modules["inverse_kinematics_pino"].PoseSolver
    orig_solve = PoseSolver.solve
    cases = []
    cur = {"frame": 0}

    def solve(self):
        with gg.LsqRecorder(m.ik) as rec:
            param, pose = orig_solve(self)
        cases.append(dict(frame=cur["frame"], cold=self.init_pose is None,
                          init=None if self.init_pose is None else (np.array(self.init_pose.root), np.array(self.init_pose.euler_angles),
                                                                    np.array(self.init_pose.bone_lens)),
                          poses=[np.array(p[:17]) for p in self.cam_poses_2d], projs=[np.array(p) for p in self.cam_projs],
                          out=(np.array(param.root), np.array(param.euler_angles), np.array(param.bone_lens)),
                          joints=np.array(pose.keypoints),
                          stage=[(r.cost, r.nfev, r.status) for _, r, _ in rec.results]))
        return param, pose

    PoseSolver.solve = solve
    F, L = SYNTH["n_frames"], SYNTH["chain_len"]
    meta = -np.ones((F, T, 4), dtype=np.int32)
    params = np.zeros((F, T, 68))
    joints = np.full((F, T, 18, 3), np.nan)
    n_tracks = np.zeros(F, dtype=np.int32)
    n_dead = np.zeros(F, dtype=np.int32)
    n_solves = np.zeros(F, dtype=np.int32)
    solve_info = []
    t0 = time.time()
    try:
        for b in range(F // L):
            tracker = m.mc.MvTracker(skel)
            ids = {}
            for t in range(L):
                f = b * L + t
                cur["frame"] = f
                n0 = len(cases)
                tracker.update_4d(t + 1, synth_frames(m, data, calibs, f), None)
                for tl in tracker.tracklets + tracker.dead_tracklets:
                    ids.setdefault(id(tl), len(ids))
                assert len(tracker.tracklets) <= T
                for s, tl in enumerate(tracker.tracklets):
                    meta[f, s] = (ids[id(tl)], tl.state.value, tl.hits, len(tl))
                    _, par, pose3d = tl.poses[-1]      # (frame, PoseShapeParam, Pose), motion_capture.py:333,367
                    params[f, s] = np.concatenate([np.ravel(par.root), np.ravel(par.euler_angles), np.ravel(par.bone_lens)])
                    joints[f, s] = pose3d.keypoints
                n_tracks[f] = len(tracker.tracklets)
                n_dead[f] = len(tracker.dead_tracklets)
                n_solves[f] = len(cases) - n0
                for c in cases[n0:]:
                    solve_info.append([f, int(c["cold"]), len(c["poses"]), c["stage"][0][1], c["stage"][0][2], c["stage"][1][1], c["stage"][1][2]])
            print(f"synthetic chain {b}: {len(cases)} solves so far, t={time.time() - t0:.0f}s", flush=True)
    finally:
        PoseSolver.solve = orig_solve
    fix = dict(seed=np.array(SYNTH["seed"]), n_frames=np.array(F), chain_len=np.array(L), n_views=np.array(C), n_people=np.array(SYNTH["n_people"]),
               kps25_checksum=np.array(float(np.abs(data["kps25"].astype(np.float64)).sum())),
               meta=meta, params=params, joints=joints, n_tracks=n_tracks, n_dead=n_dead, n_solves=n_solves,
               solve_info=np.array(solve_info), solve_joints=np.array([c["joints"] for c in cases]),
               solve_cost=np.array([c["stage"][1][0] for c in cases]))
    return fix, cases


def pick(cases, n_per_view, want_warm, rng):
    """A spread of cases over view counts; every case once as a cold start, some also from their warm init."""
    by_v = {}
    for c in cases:
        by_v.setdefault(len(c["poses"]), []).append(c)
    sel = []
    for v in sorted(by_v):
        pool = by_v[v]
        idx = rng.permutation(len(pool))[:n_per_view]
        for i in idx:
            sel.append((pool[i], None))
    warm = [c for c in cases if not c["cold"]]
    for i in rng.permutation(len(warm))[:want_warm]:
        sel.append((warm[i], warm[i]["init"]))
    return sel


def main():
    global _M
    ap = argparse.ArgumentParser()
    ap.add_argument("--shelf-frames", type=int, default=150)
    ap.add_argument("--max-nfev", type=int, default=400)
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--shelf-per-view", type=int, default=40)
    ap.add_argument("--synth-cases", type=int, default=72)
    args = ap.parse_args()
    m = ref_shim.load_modules()
    _M = m
    from pathlib import Path
    rng = np.random.default_rng(20261003)

    fix, synth_cases = run_synth_tracker(m)
    np.savez_compressed(f"{OUT}/synth_c4_tracker.npz", **fix)
    print("saved synth_c4_tracker.npz:", int(fix["n_solves"].sum()), "solves over", SYNTH["n_frames"], "frames")

    calibs = [m.mc.load_calib(Path(f"{gg.SHELF}/calibs/{c}.json")) for c in range(gg.N_CAM)]
    shelf_cases, _ = gg.run_tracker(m, calibs, args.shelf_frames, None)
    print("shelf pool:", len(shelf_cases), "solves; views:", np.bincount([len(c["poses"]) for c in shelf_cases]))

    sel = [("shelf", c, init) for c, init in pick(shelf_cases, args.shelf_per_view, 40, rng)]
    sel += [("synth", c, init) for c, init in pick(synth_cases, args.synth_cases, 24, rng)]
    sel = [s for s in sel if len(s[1]["poses"]) <= V_PAD]
    jobs = [(c["poses"], c["projs"], init, args.max_nfev) for _, c, init in sel]
    print("running", len(jobs), "converged solves on", args.procs, "processes", flush=True)
    t0 = time.time()
    with mp.get_context("fork").Pool(args.procs) as pool:
        res = pool.map(_solve_converged, jobs, chunksize=1)
    print(f"done in {time.time() - t0:.0f}s; slowest solve {max(r['secs'] for r in res):.0f}s")

    n = len(sel)
    d = dict(source=np.array([0 if s[0] == "shelf" else 1 for s in sel]), frame=np.array([s[1]["frame"] for s in sel]),
             warm_init=np.array([s[2] is not None for s in sel]), n_views=np.array([len(s[1]["poses"]) for s in sel]),
             poses=np.zeros((n, V_PAD, 17, 3)), projs=np.zeros((n, V_PAD, 3, 4)), init=np.zeros((n, 68)),
             s1_x=np.zeros((n, 57)), s1_cost=np.zeros(n), s1_nfev=np.zeros(n, int), s1_status=np.zeros(n, int),
             s2_x=np.zeros((n, 68)), s2_cost=np.zeros(n), s2_nfev=np.zeros(n, int), s2_status=np.zeros(n, int),
             joints=np.zeros((n, 18, 3)), max_nfev=np.array(args.max_nfev),
             synth_K=fix_cal(synth_cases, "K"), synth_P=fix_cal(synth_cases, "P"))
    for i, ((_, c, _), r) in enumerate(zip(sel, res)):
        v = len(c["poses"])
        d["poses"][i, :v] = np.array(c["poses"])
        d["projs"][i, :v] = np.array(c["projs"])
        d["init"][i] = np.concatenate([r["init"][0], r["init"][1].ravel(), r["init"][2]])
        d["s1_x"][i], d["s1_cost"][i], d["s1_nfev"][i], d["s1_status"][i] = r["s1"]
        d["s2_x"][i], d["s2_cost"][i], d["s2_nfev"][i], d["s2_status"][i] = r["s2"]
        d["joints"][i] = r["joints"]
    conv = (d["s1_status"] > 0) & (d["s2_status"] > 0)
    print("cases", n, "both stages converged:", int(conv.sum()), "by views:", np.bincount(d["n_views"][conv]),
          "shelf/synth:", int((conv & (d["source"] == 0)).sum()), int((conv & (d["source"] == 1)).sum()))
    np.savez_compressed(f"{OUT}/ik_converged.npz", **d)


def fix_cal(cases, what):
    from multiview_motion_capture_amd import synth
    K, Rt, P = synth.make_cameras(SYNTH["n_views"], np.random.default_rng(SYNTH["seed"]))
    return K if what == "K" else P


if __name__ == "__main__":
    main()
