"""Drop-in for the per-frame association glue of the reference's motion_capture.py: loaders
(parse_openpose_kps, load_calib, filter_bad_pose), match_spatial, parse_match_result and the
SpatialMatch / SpatialTimeMatch records.  The numeric stages run on the GPU (mv_math_util,
mv_association); only list/dict bookkeeping happens on the host, as in the reference."""
from __future__ import annotations

import json
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
    """motion_capture.py:250-272 (json branch)."""
    cpath = Path(cpath)
    if 'js' not in cpath.suffix:
        raise ValueError(f'unsupported calibration format. {cpath.name}')
    with open(str(cpath), 'r') as fh:
        js = json.load(fh)
    return Calib.from_k_rt(js["K"], js["RT"], js["imgSize"])


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
