"""Keypoint schema of the hot path (mirror of the reference's pose_def.py: same public names).

Joint vocabularies are data; they are restated here from the reference tables
(pose_def.py:8-52 KpsType, :72-96 COCO, :111-137 OpenPose-25, :181-228 BASIC_18 + parents,
:230-259 side lists) so that callers of the reference API keep working unchanged.
"""
from __future__ import annotations

import copy
from dataclasses import dataclass
from enum import Enum
from typing import Dict, Optional

import numpy as np

_KPS_NAMES = ("Nose L_Eye R_Eye L_Ear R_Ear Head_Top Head_Bottom Head Neck L_Shoulder R_Shoulder L_Elbow R_Elbow "
              "L_Wrist R_Wrist L_Hip R_Hip Mid_Hip L_Knee R_Knee L_Ankle R_Ankle Pelvis Spine L_BaseBigToe "
              "R_BaseBigToe L_BigToe R_BigToe L_SmallToe R_SmallToe L_Hand R_Hand L_Heel R_Heel Chest LowerNeck "
              "UpperNeck LowerBack UpperBack L_Clavicle R_Clavicle Root").split()
KpsType = Enum("KpsType", {name: i for i, name in enumerate(_KPS_NAMES)})
KpsType.__doc__ = "official name of each joint type (pose_def.py:8-52)"


class KpsFormat(Enum):
    COCO = 0
    OPENPOSE_25 = 1
    SMPLX_22 = 2
    BASIC_18 = 3


@dataclass
class Pose:
    pose_type: KpsFormat
    keypoints: np.ndarray
    keypoints_score: Optional[np.ndarray]
    box: Optional[np.ndarray]

    def to_kps_array(self):
        return np.concatenate([self.keypoints, self.keypoints_score.reshape((-1, 1))], axis=1)


def _types(names):
    return [KpsType[n] for n in names.split()]


_COCO = _types("Nose L_Eye R_Eye L_Ear R_Ear L_Shoulder R_Shoulder L_Elbow R_Elbow L_Wrist R_Wrist L_Hip R_Hip "
               "L_Knee R_Knee L_Ankle R_Ankle")
_OPENPOSE_25 = _types("Nose Neck R_Shoulder R_Elbow R_Wrist L_Shoulder L_Elbow L_Wrist Mid_Hip R_Hip R_Knee R_Ankle "
                      "L_Hip L_Knee L_Ankle R_Eye L_Eye R_Ear L_Ear L_BigToe L_SmallToe L_Heel R_BigToe R_SmallToe "
                      "R_Heel")
_BASIC_18 = _types("Mid_Hip L_Hip L_Knee L_Ankle R_Hip R_Knee R_Ankle Spine Neck L_Shoulder L_Elbow L_Wrist "
                   "R_Shoulder R_Elbow R_Wrist Nose L_Ear R_Ear")
_BASIC_18_PARENT_NAMES = ("Mid_Hip Mid_Hip L_Hip L_Knee Mid_Hip R_Hip R_Knee Mid_Hip Spine Neck L_Shoulder L_Elbow "
                          "Neck R_Shoulder R_Elbow Neck Nose Nose").split()
_ORDERS = {KpsFormat.COCO: _COCO, KpsFormat.OPENPOSE_25: _OPENPOSE_25, KpsFormat.BASIC_18: _BASIC_18}
_INDEX = {fmt: {j: i for i, j in enumerate(order)} for fmt, order in _ORDERS.items()}
_BASIC_18_PARENTS_Index = [(-1 if KpsType[p] == j else _INDEX[KpsFormat.BASIC_18][KpsType[p]])
                           for j, p in zip(_BASIC_18, _BASIC_18_PARENT_NAMES)]
_L_Side_Joints = _types("L_Hip L_Knee L_Ankle L_Shoulder L_Elbow L_Wrist L_Ear")
_R_Side_Joints = _types("R_Hip R_Knee R_Ankle R_Shoulder R_Elbow R_Wrist R_Ear")
_M_Side_Joints = _types("Mid_Hip Spine Neck Nose")


def get_kps_order(p_type):
    if p_type not in _ORDERS:
        raise ValueError('get_kps_index')
    return _ORDERS[p_type]


def get_kps_index(p_type) -> Dict[KpsType, int]:
    if p_type not in _INDEX:
        raise ValueError('get_kps_index')
    return copy.copy(_INDEX[p_type])


def get_parent_index(p_type):
    if p_type != KpsFormat.BASIC_18:
        raise ValueError(f'get_parent_index: {p_type}')
    return copy.copy(_BASIC_18_PARENTS_Index)


def get_pose_bones_index(p_type):
    if p_type != KpsFormat.BASIC_18:
        raise ValueError(f'get_pose_bones_index: {p_type}')
    return [(k, p) for k, p in enumerate(_BASIC_18_PARENTS_Index) if p >= 0]


def get_sides_joints(p_type):
    if p_type != KpsFormat.BASIC_18:
        raise ValueError(f'get_sides_joints {p_type}')
    return list(_L_Side_Joints), list(_R_Side_Joints), list(_M_Side_Joints)


def get_sides_joint_idxs(p_type):
    l, r, m = get_sides_joints(p_type)
    idx = _INDEX[KpsFormat.BASIC_18]
    return [idx[j] for j in l], [idx[j] for j in r], [idx[j] for j in m]


def get_joint_side(jnt_type: KpsType):
    name = jnt_type.name
    return 'left' if name.startswith('L_') else 'right' if name.startswith('R') else 'mid'


def get_flip_joint(jnt_type: KpsType):
    side = get_joint_side(jnt_type)
    if side == 'left':
        return KpsType[jnt_type.name.replace('L_', 'R_')]
    if side == 'right':
        return KpsType[jnt_type.name.replace('R_', 'L_')]
    return jnt_type


def conversion_openpose_25_to_coco(poses_openpose):
    """(25, ch) -> (17, ch) gather (pose_def.py:262-270)."""
    src = [_INDEX[KpsFormat.OPENPOSE_25][j] for j in _COCO]
    return np.ascontiguousarray(np.asarray(poses_openpose)[src, :])


def get_common_kps_idxs_1(src_kps_idx_map: Dict[KpsType, int], dst_kps_idx_map: Dict[KpsType, int]):
    pairs = [(i, dst_kps_idx_map[k]) for k, i in src_kps_idx_map.items() if k in dst_kps_idx_map]
    return [a for a, _ in pairs], [b for _, b in pairs]


def get_common_kps_idxs(src_p_type, dst_p_type):
    dst = get_kps_index(dst_p_type)
    pairs = [(i, dst[j]) for i, j in enumerate(get_kps_order(src_p_type)) if j in dst]
    return [a for a, _ in pairs], [b for _, b in pairs]


def map_to_common_keypoints(pose_0: Pose, pose_1: Pose):
    i0, i1 = get_common_kps_idxs(pose_0.pose_type, pose_1.pose_type)
    return pose_0.to_kps_array()[i0, :], pose_1.to_kps_array()[i1, :]
