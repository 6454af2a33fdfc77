"""Generate the golden vectors under tests/golden by RUNNING THE REFERENCE.

TEST INFRASTRUCTURE (build container only).  Imports the reference's own
modules from /root/reference/src through ``oracle/ref_shim.py`` and records
inputs + outputs of the hot-path functions.  Only data is written (inputs and
expected outputs as .npz); no reference source travels.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [--skip-tracker]

Fixtures written:
  fk_known_answers.npz  725 (param, joints) pairs shipped by the reference in
                        data/shelf/tracklets/traclets.pkl
  shelf_inputs.npz      OpenPose-25 keypoints of data/shelf/kps_opn (frames
                        0..300, 5 views, padded) + the 5 calibrations
  shelf_spatial.npz     per selected frame: match_spatial internals (F, D, S,
                        X_bin, match_mat, ALS iterations, clusters) + DLT
  ik_cases.npz          PoseSolver.solve() cold + warm cases with the
                        least_squares results of both stages
  shelf_tracker.npz     MvTracker.update_4d over frames 1..N: per-frame
                        association + tracker state + every IK solve
"""
import argparse
import json
import os
import pickle
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
SHELF = "/root/reference/data/shelf"
SPATIAL_FRAMES = [1, 50, 100, 131, 150, 200, 220, 295, 300]
N_CAM = 5
P_MAX = 8


def load_shelf_raw():
    """(F,C,P_MAX,25,3) f64 padded with zeros, counts (F,C)."""
    n_frames = 301
    kps = np.zeros((n_frames, N_CAM, P_MAX, 25, 3))
    cnt = np.zeros((n_frames, N_CAM), dtype=np.int32)
    for c in range(N_CAM):
        for f in range(n_frames):
            with open(f"{SHELF}/kps_opn/{c}/{c}_{f:012d}_keypoints.json") as fh:
                people = json.load(fh)["people"]
            assert len(people) <= P_MAX
            cnt[f, c] = len(people)
            for p, person in enumerate(people):
                kps[f, c, p] = np.array(person["pose_keypoints_2d"]).reshape(25, 3)
    Ks, Rts = [], []
    for c in range(N_CAM):
        with open(f"{SHELF}/calibs/{c}.json") as fh:
            js = json.load(fh)
        Ks.append(np.array(js["K"]).reshape(3, 3))
        Rts.append(np.array(js["RT"]).reshape(3, 4))
    return kps, cnt, np.array(Ks), np.array(Rts)


def ref_frames(m, frame_idx, calibs):
    """The reference's own loader + filter for one frame (motion_capture.py:974-984,1023-1043)."""
    from pathlib import Path
    frames = []
    for c in range(N_CAM):
        poses = m.mc.parse_openpose_kps(Path(f"{SHELF}/kps_opn/{c}/{c}_{frame_idx:012d}_keypoints.json"))
        fd = m.mc.FrameData(frame_idx, poses, calibs[c], view_id=c + 1)
        frames.append(m.mc.filter_bad_pose(fd, 0.01, 4, 5))
    return frames


class InvCounter:
    """Counts np.linalg.inv calls to recover match_als' iteration count (2 per iteration)."""

    def __init__(self):
        self.n = 0
        self._orig = np.linalg.inv

    def __enter__(self):
        def inv(a):
            self.n += 1
            return self._orig(a)
        np.linalg.inv = inv
        return self

    def __exit__(self, *a):
        np.linalg.inv = self._orig


def gen_fk(m):
    import types
    main = sys.modules["__main__"]
    for name in ("MvTracklet", "TrackState"):
        if not hasattr(main, name):
            setattr(main, name, getattr(m.mc, name))
    with open(f"{SHELF}/tracklets/traclets.pkl", "rb") as fh:
        data = pickle.load(fh)
    tlets = data["tracklets"]
    roots, eulers, blens, joints = [], [], [], []
    skel = tlets[0].skel
    for t in tlets:
        for entry in t.poses:
            param, pose = entry[-2], entry[-1]
            roots.append(param.root)
            eulers.append(param.euler_angles)
            blens.append(param.bone_lens)
            joints.append(pose.keypoints)
    roots, eulers, blens, joints = map(np.array, (roots, eulers, blens, joints))
    # run the reference's *current* FK on them (identity bone-length map: old schema holds 18 lengths)
    cur = m.ik.load_skeleton()
    fk_skel = m.ik.Skeleton(ref_joint_euler_angles=cur.ref_joint_euler_angles,
                            ref_bone_dirs=np.array(skel.ref_bone_dirs),
                            ref_side_bone_lens=np.zeros(18),
                            ref_side_to_full_bone_lens_map=list(range(18)),
                            n_joints=18, joint_parents=cur.joint_parents, kps_format=cur.kps_format)
    err = 0.0
    for i in range(len(roots)):
        pos, _ = m.ik.foward_kinematics(fk_skel, m.ik.PoseShapeParam(roots[i], eulers[i], blens[i]))
        err = max(err, np.abs(pos - joints[i]).max())
    print("FK known answers:", len(roots), "max |ref FK - pickled joints| =", err)
    np.savez_compressed(f"{OUT}/fk_known_answers.npz", root=roots, euler=eulers, blens_full=blens,
                        joints=joints, bone_dirs=np.array(skel.ref_bone_dirs),
                        parents=np.array(cur.joint_parents),
                        # the CURRENT load_skeleton() constants (the pickle holds an older skeleton)
                        cur_bone_dirs=np.array(cur.ref_bone_dirs), cur_side_lens=np.array(cur.ref_side_bone_lens),
                        cur_side_map=np.array(cur.ref_side_to_full_bone_lens_map))


def gen_spatial(m, calibs):
    out = {}
    for fi in SPATIAL_FRAMES:
        frames = ref_frames(m, fi, calibs)
        poses_by_view = [[frames[c].poses[k] for k in frames[c].poses.keys()] for c in range(N_CAM)]
        pts, dim = [], [0]
        for pv in poses_by_view:
            dim.append(dim[-1] + len(pv))
            pts += [p.keypoints for p in pv]
        pts = np.array(pts)
        scores = np.array([p.keypoints_score for pv in poses_by_view for p in pv])
        F = m.mu.calc_pairwise_f_mats([fr.calib for fr in frames])
        D, S = m.mu.geometry_affinity(pts, F, dim)
        with InvCounter() as ic:
            match_mat, x_bin = m.assoc.match_als(S, dim)
        clusters = m.mc.parse_match_result(match_mat, S.shape[0], dim)
        k = f"f{fi}_"
        out[k + "points"] = pts
        out[k + "scores"] = scores
        out[k + "dim"] = np.array(dim)
        out[k + "F"] = F
        out[k + "D"] = D
        out[k + "S"] = S
        out[k + "x_bin"] = x_bin
        out[k + "match_mat"] = match_mat.astype(np.uint8)
        out[k + "als_iters"] = np.array(ic.n // 2)
        out[k + "n_clusters"] = np.array(len(clusters))
        for ci, cl in enumerate(clusters):
            out[k + f"cl{ci}"] = np.array(cl)
            if len(cl) < 2:  # update_4d only solves clusters with >= 2 members (motion_capture.py:940)
                continue
            projs = [frames[g].calib.P for g, _, _ in cl]
            grps = [np.concatenate([poses_by_view[g][l].keypoints, poses_by_view[g][l].keypoints_score], axis=1)
                    for g, l, _ in cl]
            out[k + f"cl{ci}_dlt"] = m.mu.triangulate_point_groups_from_multiple_views_linear(
                np.array(projs), grps, 0.01, False)
            out[k + f"cl{ci}_dlt_post"] = m.mu.triangulate_point_groups_from_multiple_views_linear(
                np.array(projs), grps, 0.01, True)
        print("spatial frame", fi, "n =", S.shape[0], "iters =", ic.n // 2, "clusters =", len(clusters))
    np.savez_compressed(f"{OUT}/shelf_spatial.npz", **out)


class LsqRecorder:
    """Wraps the reference module's ``least_squares`` symbol to keep each OptimizeResult."""

    def __init__(self, mod):
        self.mod = mod
        self.orig = mod.least_squares
        self.results = []

    def __enter__(self):
        def wrapped(fun, x0, **kw):
            r = self.orig(fun, x0, **kw)
            self.results.append((np.array(x0, dtype=float).copy(), r, kw.get("max_nfev")))
            return r
        self.mod.least_squares = wrapped
        return self

    def __exit__(self, *a):
        self.mod.least_squares = self.orig


def run_tracker(m, calibs, n_frames, want_ik_frames):
    """MvTracker.update_4d over Shelf frames 1..n_frames with every IK solve recorded."""
    skel = m.ik.load_skeleton()
    tracker = m.mc.MvTracker(skel)
    ik_cases, frames_log = [], {}
    PoseSolver = sys.modules["inverse_kinematics_pino"].PoseSolver
    orig_solve = PoseSolver.solve

    cur = {"frame": 0}

    def solve(self):
        with LsqRecorder(m.ik) as rec_ik, LsqRecorder(m.mu) as rec_tr:
            param, pose = orig_solve(self)
        ik_cases.append(dict(
            frame=cur["frame"], cold=self.init_pose is None,
            init=None if self.init_pose is None else (np.array(self.init_pose.root), np.array(self.init_pose.euler_angles),
                                                      np.array(self.init_pose.bone_lens)),
            poses=[np.array(p[:17]) for p in self.cam_poses_2d],  # 17 real + 1 synthetic row
            projs=[np.array(p) for p in self.cam_projs],
            out=(np.array(param.root), np.array(param.euler_angles), np.array(param.bone_lens)),
            joints=np.array(pose.keypoints),
            stage=[(x0, r.x.copy(), r.cost, r.nfev, r.njev, r.status, mx) for x0, r, mx in rec_ik.results],
            tri=[(x0, r.x.copy(), r.cost, r.nfev, r.status) for x0, r, mx in rec_tr.results]))
        return param, pose

    PoseSolver.solve = solve
    ids = {}
    t0 = time.time()
    for fi in range(1, n_frames + 1):
        cur["frame"] = fi
        frames = ref_frames(m, fi, calibs)
        alive_before = [t for t in tracker.tracklets if not t.is_dead()]
        n_before = len(ik_cases)
        tracker.update_4d(fi, frames, None)
        for t in tracker.tracklets + tracker.dead_tracklets:
            ids.setdefault(id(t), len(ids))
        frames_log[fi] = dict(
            n_alive_before=len(alive_before),
            alive_after=[(ids[id(t)], t.state.value, t.hits, len(t)) for t in tracker.tracklets],
            n_dead=len(tracker.dead_tracklets),
            n_solves=len(ik_cases) - n_before,
            counts=[len(f.poses) for f in frames])
        if fi % 20 == 0:
            print(f"tracker frame {fi}: alive={len(tracker.tracklets)} dead={len(tracker.dead_tracklets)} "
                  f"t={time.time() - t0:.0f}s", flush=True)
    PoseSolver.solve = orig_solve
    return ik_cases, frames_log


def save_ik_cases(cases, path, max_cases):
    """Pad to V=5 views and store a bounded number of cases (all cold ones first)."""
    cold = [c for c in cases if c["cold"]]
    warm = [c for c in cases if not c["cold"]]
    step = max(1, len(warm) // max(1, max_cases - min(len(cold), max_cases // 2)))
    sel = cold[:max_cases // 2] + warm[::step]
    sel = sel[:max_cases]
    n = len(sel)
    d = dict(
        frame=np.array([c["frame"] for c in sel]),
        cold=np.array([c["cold"] for c in sel]),
        n_views=np.array([len(c["poses"]) for c in sel]),
        poses=np.zeros((n, N_CAM + 1, 17, 3)), projs=np.zeros((n, N_CAM + 1, 3, 4)),
        init_root=np.zeros((n, 3)), init_euler=np.zeros((n, 18, 3)), init_blens=np.zeros((n, 11)),
        out_root=np.zeros((n, 3)), out_euler=np.zeros((n, 18, 3)), out_blens=np.zeros((n, 11)),
        joints=np.zeros((n, 18, 3)),
        s1_x0=np.zeros((n, 57)), s1_x=np.zeros((n, 57)), s1_cost=np.zeros(n), s1_nfev=np.zeros(n, int),
        s1_njev=np.zeros(n, int), s1_status=np.zeros(n, int),
        s2_x0=np.zeros((n, 68)), s2_x=np.zeros((n, 68)), s2_cost=np.zeros(n), s2_nfev=np.zeros(n, int),
        s2_njev=np.zeros(n, int), s2_status=np.zeros(n, int),
        tri_x0=np.zeros((n, 54)), tri_x=np.zeros((n, 54)), tri_nfev=np.zeros(n, int), tri_status=np.zeros(n, int))
    for i, c in enumerate(sel):
        v = len(c["poses"])
        assert v <= N_CAM + 1
        d["poses"][i, :v] = np.array(c["poses"])
        d["projs"][i, :v] = np.array(c["projs"])
        if c["init"] is not None:
            d["init_root"][i], d["init_euler"][i], d["init_blens"][i] = c["init"]
        d["out_root"][i], d["out_euler"][i], d["out_blens"][i] = c["out"]
        d["joints"][i] = c["joints"]
        (x0, x, cost, nfev, njev, st, mx) = c["stage"][0]
        d["s1_x0"][i], d["s1_x"][i], d["s1_cost"][i], d["s1_nfev"][i], d["s1_njev"][i], d["s1_status"][i] = x0, x, cost, nfev, njev, st
        (x0, x, cost, nfev, njev, st, mx) = c["stage"][1]
        d["s2_x0"][i], d["s2_x"][i], d["s2_cost"][i], d["s2_nfev"][i], d["s2_njev"][i], d["s2_status"][i] = x0, x, cost, nfev, njev, st
        if c["tri"]:
            (x0, x, cost, nfev, st) = c["tri"][0]
            d["tri_x0"][i], d["tri_x"][i], d["tri_nfev"][i], d["tri_status"][i] = x0, x, nfev, st
    np.savez_compressed(path, **d)
    print("saved", n, "IK cases (", int(d["cold"].sum()), "cold ) ->", path)


def save_tracker_log(frames_log, cases, path):
    fr = sorted(frames_log)
    amax = max(len(frames_log[f]["alive_after"]) for f in fr)
    alive = -np.ones((len(fr), amax, 4), dtype=np.int32)
    for i, f in enumerate(fr):
        for j, rec in enumerate(frames_log[f]["alive_after"]):
            alive[i, j] = rec
    # per solve: frame, cold flag, root out (3) -- a compact trajectory check
    solves = np.array([[c["frame"], int(c["cold"]), len(c["poses"])] + list(c["out"][0]) for c in cases])
    joints = np.array([c["joints"] for c in cases])
    np.savez_compressed(path, frames=np.array(fr), alive_after=alive,
                        n_alive_before=np.array([frames_log[f]["n_alive_before"] for f in fr]),
                        n_dead=np.array([frames_log[f]["n_dead"] for f in fr]),
                        n_solves=np.array([frames_log[f]["n_solves"] for f in fr]),
                        counts=np.array([frames_log[f]["counts"] for f in fr]),
                        solves=solves, solve_joints=joints)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-tracker", action="store_true")
    ap.add_argument("--tracker-frames", type=int, default=300)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    m = ref_shim.load_modules()
    from pathlib import Path
    calibs = [m.mc.load_calib(Path(f"{SHELF}/calibs/{c}.json")) for c in range(N_CAM)]

    if not args.only or "inputs" in args.only:
        kps, cnt, Ks, Rts = load_shelf_raw()
        np.savez_compressed(f"{OUT}/shelf_inputs.npz", kps25=kps, counts=cnt, K=Ks, Rt=Rts,
                            P=np.array([c.P for c in calibs]))
        print("shelf inputs", kps.shape, "max people/view", cnt.max())
    if not args.only or "fk" in args.only:
        gen_fk(m)
    if not args.only or "spatial" in args.only:
        gen_spatial(m, calibs)
    if not args.skip_tracker and (not args.only or "tracker" in args.only):
        cases, log = run_tracker(m, calibs, args.tracker_frames, None)
        save_ik_cases(cases, f"{OUT}/ik_cases.npz", 64)
        save_tracker_log(log, cases, f"{OUT}/shelf_tracker.npz")


if __name__ == "__main__":
    main()
