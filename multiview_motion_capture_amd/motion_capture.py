"""Drop-in for the per-frame association glue of the reference's motion_capture.py: loaders
(parse_openpose_kps, load_calib, filter_bad_pose), match_spatial, parse_match_result and the
SpatialMatch / SpatialTimeMatch records.  The numeric stages run on the GPU (mv_math_util,
mv_association); only list/dict bookkeeping happens on the host, as in the reference."""
from __future__ import annotations

import json
import os
import pickle
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np

from .common import Calib, FrameData
from .mv_association import match_als
from .mv_math_util import calc_pairwise_f_mats, geometry_affinity
from .pose_def import KpsFormat, Pose, conversion_openpose_25_to_coco


@dataclass
class SpatialMatch:
    view_idxs: List[int] = field(default_factory=list)
    pose_ids: List[int] = field(default_factory=list)
    cost_matrix_idxs: List[int] = field(default_factory=list)

    def __len__(self):
        return len(self.view_idxs)


@dataclass
class SpatialTimeMatch:
    spatial_time_matches: Dict[int, SpatialMatch]
    spatial_matches: List[SpatialMatch]
    tlet_matrix_idxs: Dict[int, int] = field(default_factory=dict)
    view_pose_matrix_idxs: Dict[int, list] = field(default_factory=dict)
    dst_mat: Optional[np.ndarray] = None
    sim_mat: Optional[np.ndarray] = None
    match_mat: Optional[np.ndarray] = None


def load_calib(cpath: Path) -> Calib:
    """motion_capture.py:250-272: a calibration file, JSON ("K", "RT", "imgSize") or pickle ("K", "R", "t")."""
    cpath = Path(cpath)
    if 'pkl' in cpath.suffix:
        with open(str(cpath), 'rb') as fh:
            data = pickle.load(fh)
        rt = np.concatenate([np.array(data["R"]).reshape((3, 3)), np.array(data["t"]).reshape((3, 1))], axis=1)
        return Calib.from_k_rt(data["K"], rt, (1920, 1080))
    if 'js' in cpath.suffix:
        with open(str(cpath), 'r') as fh:
            js = json.load(fh)
        return Calib.from_k_rt(js["K"], js["RT"], js["imgSize"])
    raise ValueError(f'unsupported calibration format. {cpath.name}')


def parse_openpose_kps(js_path: Path) -> Dict[int, Pose]:
    """motion_capture.py:974-984: OpenPose JSON -> {person id: COCO-17 Pose}."""
    with open(js_path, 'rt') as fh:
        people = json.load(fh)["people"]
    poses = {}
    for p_id, person in enumerate(people):
        coco = conversion_openpose_25_to_coco(np.array(person["pose_keypoints_2d"]).reshape((-1, 3)))
        poses[p_id] = Pose(KpsFormat.COCO, keypoints=coco[:, :2], keypoints_score=coco[:, -1][:, np.newaxis], box=None)
    return poses


def filter_bad_pose(smt_frm: FrameData, min_valid_kps_score, n_min_valid_kps, min_valib_bb_size) -> FrameData:
    """motion_capture.py:1023-1043 on the FrameData dict (the batched path does this inside mvmc_ingest)."""
    bad = []
    for p_id, pose in smt_frm.poses.items():
        ok = (pose.keypoints_score > min_valid_kps_score).flatten()
        if ok.sum() < n_min_valid_kps:
            bad.append(p_id)
            continue
        xy = pose.keypoints[ok, :2]
        if np.any((xy.max(axis=0) - xy.min(axis=0)) < min_valib_bb_size):
            bad.append(p_id)
    for p_id in bad:
        del smt_frm.poses[p_id]
    return smt_frm


def parse_match_result(match_mat_: np.ndarray, n, dims_group):
    """motion_capture.py:417-446 -> clusters, each a list of (group, local index, global index)."""
    mm = np.asarray(match_mat_).astype(np.float64)
    keep = np.nonzero(mm.sum(axis=0) > 1.9)[0]
    if keep.size == 0:
        raise RuntimeError("parse_match_result: no cluster with at least two members")  # torch reshape raises there
    member = mm[:, keep] > 0.9
    clusters = [[] for _ in keep]
    for row in range(n):
        if member[row].any():
            clusters[int(np.argmax(member[row]))].append(row)
    dg = np.asarray(dims_group)
    out = []
    for rows in clusters:
        cur = []
        for idx in rows:
            grp = int(np.nonzero(dg <= idx)[0][-1])
            cur.append((grp, int(idx - dg[grp]), int(idx)))
        if cur:
            out.append(cur)
    return out


def match_spatial(frames: List[FrameData]) -> SpatialTimeMatch:
    """motion_capture.py:597-631."""
    ids = [list(frm.poses.keys()) for frm in frames]
    points, dim_groups = [], [0]
    for frm, pid in zip(frames, ids):
        dim_groups.append(dim_groups[-1] + len(pid))
        points += [frm.poses[p].keypoints for p in pid]
    points = np.array(points)
    f_mats = calc_pairwise_f_mats([frm.calib for frm in frames])
    dst_mat, s_mat = geometry_affinity(points, f_mats, dim_groups)
    match_mat, _ = match_als(s_mat, dim_groups)
    out = SpatialTimeMatch({}, [])
    for cluster in parse_match_result(match_mat, s_mat.shape[0], dim_groups):
        m = SpatialMatch([], [])
        for cam_idx, p_idx, _ in cluster:
            m.view_idxs.append(cam_idx)
            m.pose_ids.append(ids[cam_idx][p_idx])
        out.spatial_matches.append(m)
    out.dst_mat, out.sim_mat = dst_mat, s_mat
    return out


# ------------------------------------------------------------------------------------------------
# temporal layer: reprojection_error, match_spatial_time, MvTracklet / MvTracker
# ------------------------------------------------------------------------------------------------
from enum import Enum  # noqa: E402

import torch  # noqa: E402

from . import device as _dev  # noqa: E402
from .inverse_kinematics import PoseShapeParam, Skeleton, load_skeleton  # noqa: E402,F401
from .mv_math_util import _d, _pair_graph  # noqa: E402


class TrackState(Enum):
    Tentative = 1
    Confirmed = 2
    Dead = 3


def reprojection_error(p_3d: Pose, p_2d: Pose, calib: Calib, min_valid_kps_score=0.05,
                       invalid_default_error=np.nan):
    """motion_capture.py:403-414 (BASIC_18 3-D pose against a COCO 2-D pose)."""
    k2 = np.concatenate([p_2d.keypoints, np.asarray(p_2d.keypoints_score).reshape(-1, 1)], axis=1)
    e = float(_pair_graph(p_3d.keypoints, [k2], [calib.P], min_valid_kps_score)[0, 1])
    return invalid_default_error if np.isnan(e) else e


def match_spatial_time(tlets, frames: List[FrameData], pixel_error_threshold=None) -> SpatialTimeMatch:
    """motion_capture.py:634-826: tracklets (their last 3-D pose) and the frame's 2-D poses in one graph -- epipolar error
    between 2-D poses of different views, reprojection error between a tracklet and a 2-D pose, NaN elsewhere -> affinity ->
    match_als -> clusters; a cluster with a tracklet becomes spatial_time_matches[tracklet index], a 2-D-only cluster a new
    spatial match; one pose per view, the first wins.  ``pixel_error_threshold`` is unused, as in the reference."""
    d = _d()
    C = len(frames)
    T = max(len(tlets), 1)
    ids = [list(frm.poses.keys()) for frm in frames]
    P = max(max((len(i) for i in ids), default=0), 1)
    kps = np.zeros((1, C, P, 17, 3))
    cnt = np.zeros((1, C), dtype=np.int32)
    for c, frm in enumerate(frames):
        cnt[0, c] = len(ids[c])
        for k, p_id in enumerate(ids[c]):
            kps[0, c, k, :, :2] = frm.poses[p_id].keypoints
            kps[0, c, k, :, 2] = np.asarray(frm.poses[p_id].keypoints_score).ravel()
    tj = np.zeros((1, T, 18, 3))
    for k, t in enumerate(tlets):
        tj[0, k] = np.asarray(t.last_pose_3d.keypoints)
    Pm = torch.as_tensor(np.array([f.calib.P for f in frames], np.float64), device=d).contiguous()
    W, D, gc = _dev.st_affinity(torch.as_tensor(kps, device=d), torch.as_tensor(cnt, device=d),
                                torch.zeros(1, dtype=torch.int32, device=d), torch.as_tensor(tj, device=d),
                                torch.tensor([len(tlets)], dtype=torch.int32, device=d), Pm, _dev.fmats_from_projections(Pm),
                                want_D=True)
    dim_groups = np.concatenate([[0], np.cumsum(gc[0].cpu().numpy())]).tolist()
    n = dim_groups[-1]
    # compact node order = [tracklets | poses by view] with the padded slots removed
    keep = list(range(len(tlets)))
    for c in range(C):
        keep += [T + c * P + k for k in range(cnt[0, c])]
    s_mat = W[0].cpu().numpy()[:n, :n]
    dst_mat = D[0].cpu().numpy()[:n, :n]
    match_mat, x_bin = match_als(s_mat, dim_groups)
    out = SpatialTimeMatch({}, [])
    nt = len(tlets)
    for cluster in parse_match_result(match_mat, n, dim_groups):
        tracklet_idx = next((gi for _, _, gi in cluster if gi < nt), -1)
        m = SpatialMatch([], [])
        for grp, loc, gi in cluster:
            if gi < nt:
                continue
            view_idx = grp - 1
            if view_idx in m.view_idxs:
                continue   # more than one pose of a view in the cluster: the first wins (:785-787, :803-805)
            m.view_idxs.append(view_idx)
            m.pose_ids.append(ids[view_idx][loc])
            m.cost_matrix_idxs.append(gi)
        if len(m) > 0:
            if tracklet_idx >= 0:
                out.spatial_time_matches[tracklet_idx] = m
            else:
                out.spatial_matches.append(m)
    for c in range(C):
        out.view_pose_matrix_idxs[c] = [(p_id, dim_groups[c + 1] + k) for k, p_id in enumerate(ids[c])]
    out.dst_mat, out.sim_mat, out.match_mat = dst_mat, s_mat, x_bin
    return out


def associate_tracking(tlets, frames: List[FrameData], min_pixel_error_hard_threshold=None) -> SpatialTimeMatch:
    """motion_capture.py:829-835."""
    return match_spatial_time(tlets, frames, min_pixel_error_hard_threshold) if tlets else match_spatial(frames)


class MvTracklet:
    """Host-side record of one tracklet (motion_capture.py:312-400); the numbers come from the device."""

    def __init__(self, tid, frm_idx, pparam, pose):
        self.track_id = tid
        self.frame_idxs = [frm_idx]
        self.poses = [(frm_idx, pparam, pose)]
        self.state = TrackState.Tentative
        self.hits = 1
        self.time_since_update = 0

    @property
    def last_pose_3d(self):
        return self.poses[-1][-1]

    def __len__(self):
        return len(self.frame_idxs)

    def is_tentative(self):
        return self.state == TrackState.Tentative

    def is_confirmed(self):
        return self.state == TrackState.Confirmed

    def is_dead(self):
        return self.state == TrackState.Dead


class MvTracker:
    """MvTracker (motion_capture.py:840-963): update_4d runs association (match_spatial /
    match_spatial_time), the IK solves and the track bookkeeping on the GPU for one sequence."""

    def __init__(self, skel: Skeleton = None, p_max: int = 8, t_max: int = 8):
        self.skeleton = skel or load_skeleton()
        self.tracklets: List[MvTracklet] = []
        self.dead_tracklets: List[MvTracklet] = []
        self._p_max, self._t_max = p_max, t_max
        self._chain = None
        self._by_id = {}
        self._calm = 0      # consecutive frames a widened tracker has stayed below t_max (update_4d narrows it back after eight)

    def _ensure(self, d_frames):
        if self._chain is None:
            from .pipeline import HotPath
            from .tracker import ChainTracker
            hp = HotPath(np.array([f.calib.K for f in d_frames]), np.array([f.calib.Rt for f in d_frames]), device=_d())
            self._chain = ChainTracker(hp, 1, self._p_max, self._t_max)

    def update_4d(self, frm_idx: int, d_frames: List[FrameData], debug_view_imgs=None):
        self._ensure(d_frames)
        ch = self._chain
        C, P = len(d_frames), self._p_max
        # (the frame's inputs are written straight into the tracker's pinned staging buffer: one asynchronous copy to the device)
        inp = ch.frame_inputs()
        kps, cnt = inp["kps_np"], inp["cnt_np"]
        kps.fill(0.0)
        cnt.fill(0)
        for c, frm in enumerate(d_frames):
            if len(frm.poses) > P:
                raise ValueError(f"update_4d: more than p_max={P} people in view {c}")
            for k, pose in enumerate(frm.poses.values()):
                kps[0, c, k, :, :2] = pose.keypoints
                kps[0, c, k, :, 2] = np.asarray(pose.keypoints_score).ravel()
            cnt[0, c] = len(frm.poses)
        ch.upload_inputs()
        k_d, c_d = inp["kps_d"], inp["cnt_d"]
        n_nodes = int(cnt.sum())
        # the state in front of the frame, should the frame have to be redone: the host mirror the last frame's read_back() left
        # (restore_previous), or -- first frame of a tracker, or after a frame that did not end in read_back -- a device snapshot
        snap = None if ch.has_previous else ch.snapshot()
        # one launch per frame (the chain kernel) when the frame's graph fits its association variants, seven otherwise
        one_launch = ch.fused_ok and (C * P > 40 or (n_nodes <= 24 and n_nodes + len(self.tracklets) <= 32))
        if one_launch:
            ch.step_fused(k_d, c_d, fold_void=False)      # (read_back() below reads the launch's void words itself)
        else:
            ch.step(k_d, c_d)
        try:
            host = ch.read_back()      # the frame's verdict AND its tables: one transfer, one synchronisation
        except ValueError:
            # The reference has no capacities (motion_capture.py:417-446, :763-808).  The frame is redone from the state saved before
            # it with the widest tables the kernels hold (tracker.T_WIDE tracklet slots, the per-stage path); only a frame beyond
            # those raises, and then the tracker is left as it was before the frame.
            from .tracker import T_WIDE
            if snap is not None:
                ch.restore(snap)
            else:
                ch.restore_previous()
            if ch.T >= T_WIDE and not one_launch:
                raise          # (already the widest tables, through the per-stage path: the replay below would be the same frame again)
            wide = ch if ch.T >= T_WIDE else ch.widened(T_WIDE)
            wsnap = wide.snapshot()
            wide.step(k_d, c_d)
            try:
                host = wide.read_back()
            except ValueError:
                wide.restore(wsnap)
                raise
            self._chain = ch = wide
        # (only now, the frame having gone through: a frame that raises leaves the tracker -- host side included -- as it was)
        for t in self.tracklets:
            t.time_since_update += 1
        n = int(host["n_tracks"][0])
        meta, params, joints = host["meta"][0, :n], host["params"][0, :n], host["joints"][0, :n]   # (views of a buffer the next frame overwrites: copied below)
        if ch.T > self._t_max:
            # the crowd has thinned out: back to the tables of the SMALL layout (a wide tracker's frame is one launch too, but of the BIG
            # layout -- a 512-thread workgroup with 129 KB of LDS, ~2 x the latency of the SMALL one's frame)
            # -- once it has STAYED at or below t_max - 1 for a few frames: a scene that hovers around t_max would otherwise pay a voided
            # launch, a restore and a widened replay every other frame
            self._calm = self._calm + 1 if n <= self._t_max - 1 else 0
            if self._calm >= 8:
                self._chain = ch = ch.narrowed(self._t_max)
                self._calm = 0
        alive = []
        for k in range(n):
            tid, state, hits, length = (int(v) for v in meta[k])
            x = params[k]
            pparam = PoseShapeParam(x[:3].copy(), x[3:57].reshape(18, 3).copy(), x[57:].copy())
            pose = Pose(KpsFormat.BASIC_18, joints[k].copy(), np.ones((18, 1)), None)
            t = self._by_id.get(tid)
            if t is None:
                t = MvTracklet(tid, frm_idx, pparam, pose)
                self._by_id[tid] = t
            elif hits > t.hits:
                t.frame_idxs.append(frm_idx)
                t.poses.append((frm_idx, pparam, pose))
                t.time_since_update = 0
            t.hits, t.state = hits, TrackState(state)
            alive.append(t)
        ids = {t.track_id for t in alive}
        for t in self.tracklets:
            if t.track_id not in ids:
                t.state = TrackState.Dead
                self.dead_tracklets.append(t)
        self.tracklets = alive


# ----------------------------------------------------------------------------------------------------
# file formats of the reference's driver: OpenPose JSON directories, per-frame FrameData
# pickles, the final {"tracklets": [...]} pickle
# ----------------------------------------------------------------------------------------------------
def _openpose_layout(in_dir: Path, calib_dir: Path):
    """Camera directories sorted by name, their calibrations, and per camera the frame files sorted by the frame
    number in '<cam>_<frame>_keypoints.json' (extract_frame_data_from_openpose, motion_capture.py:987-996)."""
    in_dir, calib_dir = Path(in_dir), Path(calib_dir)
    cam_dirs = sorted([d for d in in_dir.glob('*') if d.is_dir()], key=lambda path: path.stem)
    calib_paths = {c.stem: c for c in calib_dir.glob('*.*')}
    calibs = [load_calib(calib_paths[v.stem]) for v in cam_dirs]
    cam_kps_paths = [sorted(d.glob('*.json'), key=lambda path: int(path.stem.split('_')[1])) for d in cam_dirs]
    n_frms = min(len(k) for k in cam_kps_paths)
    return calibs, cam_kps_paths, n_frms


def extract_frame_data_from_openpose(in_dir: Path, calib_dir: Path, out_data_dir: Path):
    """motion_capture.py:987-1005: one pickle per frame holding List[FrameData] (view_id = camera ordinal + 1)."""
    calibs, cam_kps_paths, n_frms = _openpose_layout(in_dir, calib_dir)
    out_data_dir = Path(out_data_dir)
    os.makedirs(out_data_dir, exist_ok=True)
    for frm_idx in range(n_frms):
        cam_poses = [parse_openpose_kps(kps_paths[frm_idx]) for kps_paths in cam_kps_paths]
        d_frames = [FrameData(frm_idx, poses, calib, view_id=v_idx + 1)
                    for v_idx, (poses, calib) in enumerate(zip(cam_poses, calibs))]
        with open(out_data_dir / f'{str(frm_idx).zfill(6)}.pkl', 'wb') as fh:
            pickle.dump(obj=d_frames, file=fh)


def load_pickle(fpath: Path, mode):
    """motion_capture.py:1008-1010."""
    with open(fpath, mode) as fh:
        return pickle.load(fh)


def load_openpose_sequence(in_dir: Path, calib_dir: Path, p_max: Optional[int] = None, frames: Optional[range] = None):
    """The same directory as one batch for the device path: (kps25 (F,C,P,25,3) f64 zero padded, counts (F,C) i32,
    calibs).  Rows are OpenPose's own (x, y, score) triples in file order; mvmc_ingest does the 25 -> 17 gather
    and filter_bad_pose."""
    calibs, cam_kps_paths, n_frms = _openpose_layout(in_dir, calib_dir)
    frames = range(n_frms) if frames is None else frames
    people = []
    for f in frames:
        row = []
        for kps_paths in cam_kps_paths:
            with open(kps_paths[f], 'rt') as fh:
                row.append([np.array(p["pose_keypoints_2d"], np.float64).reshape((-1, 3)) for p in json.load(fh)["people"]])
        people.append(row)
    most = max((len(v) for row in people for v in row), default=0)
    if p_max is None:
        p_max = max(most, 1)
    if most > p_max:
        raise ValueError(f"load_openpose_sequence: {most} people in one view, p_max = {p_max}")
    kps = np.zeros((len(people), len(cam_kps_paths), p_max, 25, 3))
    counts = np.zeros((len(people), len(cam_kps_paths)), dtype=np.int32)
    for f, row in enumerate(people):
        for c, view in enumerate(row):
            counts[f, c] = len(view)
            for k, arr in enumerate(view):
                if arr.shape != (25, 3):
                    raise ValueError(f"load_openpose_sequence: expected 25 keypoints per person, got {arr.shape[0]}")
                kps[f, c, k] = arr
    return kps, counts, calibs


def frame_data_from_batch(frm_idx: int, kps25_f: np.ndarray, counts_f: np.ndarray, calibs: List[Calib]) -> List[FrameData]:
    """One frame of a batch as the reference's List[FrameData] (what extract_frame_data_from_openpose pickles)."""
    out = []
    for v_idx, calib in enumerate(calibs):
        poses = {}
        for p_id in range(int(counts_f[v_idx])):
            coco = conversion_openpose_25_to_coco(kps25_f[v_idx, p_id])
            poses[p_id] = Pose(KpsFormat.COCO, keypoints=coco[:, :2], keypoints_score=coco[:, -1][:, np.newaxis], box=None)
        out.append(FrameData(frm_idx, poses, calib, view_id=v_idx + 1))
    return out


def run_main(video_dir: Optional[Path], pose_dir: Path, out_dir: Path, n_test: int = 300):
    """run_main (motion_capture.py:1047-1129) without the video readers (they only feed the debug drawings): per-frame
    pickles of pose_dir in frame order, filter_bad_pose(0.01, 4, 5), MvTracker.update_4d, and the tracklets -- longest
    first -- as {"tracklets": [...]} in out_dir/tracklets.pkl.  Like the reference the loop starts at the second
    file (frm_idx is incremented before the first read) and stops after n_test frames."""
    frm_pose_paths = sorted(Path(pose_dir).glob('*.pkl'), key=lambda path: int(path.stem))
    tracker = MvTracker(load_skeleton())
    n_test = min(len(frm_pose_paths), n_test)
    frm_idx = 0
    while True:
        frm_idx += 1
        if frm_idx >= len(frm_pose_paths):
            break
        d_frames = load_pickle(frm_pose_paths[frm_idx], 'rb')
        d_frames = [filter_bad_pose(frm, min_valid_kps_score=0.01, n_min_valid_kps=4, min_valib_bb_size=5) for frm in d_frames]
        tracker.update_4d(frm_idx, d_frames, debug_view_imgs=None)
        if frm_idx >= n_test:
            break
    all_tlets = sorted(tracker.tracklets + tracker.dead_tracklets, key=lambda tlet: -len(tlet))
    os.makedirs(out_dir, exist_ok=True)
    with open(f'{out_dir}/tracklets.pkl', 'wb') as fh:
        pickle.dump(file=fh, obj={"tracklets": all_tlets})
    return all_tlets
