"""Drop-in for the SciPy IK of the reference's inverse_kinematics.py: Skeleton, PoseShapeParam,
load_skeleton, foward_kinematics (sic) and PoseSolver, all backed by the gfx950 kernels."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import device as dev
from .mv_math_util import _d
from .pose_def import KpsFormat, Pose, get_kps_index, get_parent_index


@dataclass
class PoseShapeParam:
    root: np.ndarray
    euler_angles: np.ndarray
    bone_lens: np.ndarray


@dataclass
class Skeleton:
    ref_joint_euler_angles: np.ndarray
    ref_bone_dirs: np.ndarray
    ref_side_bone_lens: np.ndarray  # left side plus mid bone lengths
    ref_side_to_full_bone_lens_map: List[int]
    n_joints: int
    joint_parents: np.ndarray
    kps_format: KpsFormat

    @property
    def skel_kps_idx_map(self):
        return get_kps_index(self.kps_format)

    @property
    def bone_idxs(self):
        return [(i + 1, p) for i, p in enumerate(self.joint_parents[1:])]

    def to_full_bone_lens(self, side_blens):
        assert len(side_blens) == len(self.ref_side_bone_lens)
        return np.array([side_blens[i] for i in self.ref_side_to_full_bone_lens_map])

    def _to_c(self):
        return dev.make_skeleton(self.ref_bone_dirs, self.joint_parents, self.ref_side_to_full_bone_lens_map,
                                 len(self.ref_side_bone_lens), self.ref_side_bone_lens)


def load_skeleton() -> Skeleton:
    """inverse_kinematics.py:120-173."""
    dirs, side = dev.skeleton_arrays()
    return Skeleton(ref_joint_euler_angles=np.zeros((18, 3)), ref_bone_dirs=dirs, ref_side_bone_lens=side,
                    ref_side_to_full_bone_lens_map=[int(i) for i in dev.SKEL_SIDE_MAP],
                    joint_parents=get_parent_index(KpsFormat.BASIC_18), n_joints=18, kps_format=KpsFormat.BASIC_18)


def foward_kinematics(skel: Skeleton, param: PoseShapeParam):
    """inverse_kinematics.py:176-199 -> (g_pos (18,3), g_transforms (18,4,4))."""
    d = _d()
    root = np.zeros(3) if param.root is None else np.asarray(param.root, np.float64).ravel()
    x = np.concatenate([root, np.asarray(param.euler_angles, np.float64).ravel(),
                        np.asarray(param.bone_lens, np.float64).ravel()])
    joints, G = dev.fk(torch.as_tensor(x[None], device=d), skel._to_c(), want_G=True)
    return joints[0].cpu().numpy(), G[0].cpu().numpy()


class PoseSolver:
    """inverse_kinematics.py:351-433.  ``cam_calibs`` is accepted and ignored: the reference's driver
    passes it (pinocchio signature, motion_capture.py:326-331) although the SciPy class lacks it."""

    def __init__(self, skeleton: Skeleton, init_pose: Optional[PoseShapeParam], cam_poses_2d: List[np.ndarray],
                 cam_projs: List[np.ndarray], cam_calibs=None, obs_kps_format: KpsFormat = KpsFormat.COCO):
        if obs_kps_format != KpsFormat.COCO:
            raise ValueError("PoseSolver: COCO-17 observations expected")
        if len(cam_poses_2d) != len(cam_projs) or len(cam_poses_2d) < 2:
            raise ValueError("PoseSolver: need >= 2 views with one projection matrix each")
        self.skel = skeleton
        self.n_joints = skeleton.n_joints
        self.init_pose = init_pose
        self.cam_poses_2d = [np.asarray(p, np.float64) for p in cam_poses_2d]
        self.cam_projs = [np.asarray(p, np.float64) for p in cam_projs]
        self.obs_kps_format = obs_kps_format
        self.skel_kps_format = skeleton.kps_format
        self.last_info = None

    def solve(self) -> Tuple[PoseShapeParam, Pose]:
        d = _d()
        V = len(self.cam_poses_2d)
        kps = torch.as_tensor(np.array(self.cam_poses_2d)[None, :, None], device=d).contiguous()  # (1,V,1,17,3)
        P = torch.as_tensor(np.array(self.cam_projs), device=d).contiguous()
        mem = torch.arange(V, dtype=torch.int32, device=d)[None]
        init = cold = None
        if self.init_pose is not None:
            p = self.init_pose
            x0 = np.concatenate([np.asarray(p.root, np.float64).ravel(), np.asarray(p.euler_angles, np.float64).ravel(),
                                 np.asarray(p.bone_lens, np.float64).ravel()])
            init = torch.as_tensor(x0[None], device=d)
            cold = torch.zeros(1, dtype=torch.uint8, device=d)
        params, joints, info = dev.ik_solve(kps, P, mem, init, cold, 50, 5, self.skel._to_c())
        x = params[0].cpu().numpy()
        self.last_info = info[0].cpu().numpy()
        return (PoseShapeParam(x[:3].copy(), x[3:57].reshape(18, 3).copy(), x[57:].copy()),
                Pose(keypoints=joints[0].cpu().numpy(), keypoints_score=np.ones((18, 1)), box=None,
                     pose_type=KpsFormat.BASIC_18))


# ----------------------------------------------------------------------------------------------------
# the stages of PoseSolver.solve as the module-level functions the reference has, and the 3-D-target variants
# ----------------------------------------------------------------------------------------------------
# skeleton joint <-> observation row for the BASIC_18 skeleton against COCO-17 + mid-spine (get_common_kps_idxs_1)
SKEL_KPS_IDXS = [1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15, 16, 17]
OBS_KPS_IDXS = [11, 13, 15, 12, 14, 16, 17, 5, 7, 9, 6, 8, 10, 0, 3, 4]


def guess_mid_spine(pose_2d: np.ndarray, kps_idx_map=None):
    """inverse_kinematics.py:339-348 for a COCO-17 pose: the mean of shoulders and hips, score = product of the four."""
    pose_2d = np.asarray(pose_2d, np.float64)
    mid_shoulder = 0.5 * (pose_2d[5, :] + pose_2d[6, :])
    midhip = 0.5 * (pose_2d[11, :] + pose_2d[12, :])
    spine = 0.5 * (mid_shoulder + midhip)
    score = pose_2d[5, -1] * pose_2d[6, -1]
    score *= pose_2d[11, -1] * pose_2d[12, -1]
    return np.array([spine[0], spine[1], score])


def _check_idxs(obs_kps_idxs, skel_kps_idxs):
    if list(obs_kps_idxs) != OBS_KPS_IDXS or list(skel_kps_idxs) != SKEL_KPS_IDXS:
        raise ValueError("the device solver is built for the BASIC_18 skeleton against COCO-17 + mid-spine observations "
                         "(the index lists PoseSolver derives); other pairings are not supported")


def _param_vec(p: PoseShapeParam):
    return np.concatenate([np.asarray(p.root, np.float64).ravel(), np.asarray(p.euler_angles, np.float64).ravel(),
                           np.asarray(p.bone_lens, np.float64).ravel()])


def _stage_reproj(skel, obs_pose_2d, obs_kps_idxs, cam_projs, skel_kps_idxs, init_param, n_max_iter, stage):
    _check_idxs(obs_kps_idxs, skel_kps_idxs)
    obs = np.asarray(obs_pose_2d, np.float64)
    if obs.ndim != 3 or obs.shape[1] not in (17, 18) or obs.shape[2] != 3 or len(cam_projs) != obs.shape[0]:
        raise ValueError("obs_pose_2d: expected (V,17|18,3) with one projection matrix per view")
    if obs.shape[1] == 18:
        for v in range(obs.shape[0]):
            if not np.array_equal(obs[v, 17], guess_mid_spine(obs[v, :17])):
                raise ValueError("row 17 must be guess_mid_spine of the 17 COCO rows (the kernel forms it itself)")
    d = _d()
    V = obs.shape[0]
    kps = torch.as_tensor(np.ascontiguousarray(obs[None, :, None, :17]), device=d)  # (1,V,1,17,3)
    P = torch.as_tensor(np.array(cam_projs, np.float64), device=d).contiguous()
    mem = torch.arange(V, dtype=torch.int32, device=d)[None]
    x0 = torch.as_tensor(_param_vec(init_param)[None], device=d)
    params, _, info = dev.ik_solve_stages(x0, stage, int(n_max_iter), kps, P, mem, None, skel._to_c())
    x = params[0].cpu().numpy()
    return x, info[0].cpu().numpy()


def solve_pose_reproj(skel: Skeleton, obs_pose_2d: np.ndarray, obs_kps_idxs: List[int], cam_projs: List[np.ndarray],
                      skel_kps_idxs: List[int], init_param: PoseShapeParam, n_max_iter=5) -> PoseShapeParam:
    """inverse_kinematics.py:202-238: root and angles against the 2-D observations, bone lengths fixed."""
    x, _ = _stage_reproj(skel, obs_pose_2d, obs_kps_idxs, cam_projs, skel_kps_idxs, init_param, n_max_iter, 1)
    return PoseShapeParam(x[:3].copy(), x[3:57].reshape(18, 3).copy(), init_param.bone_lens)


def solve_pose_bone_lens_reproj(skel: Skeleton, obs_pose_2d: np.ndarray, obs_kps_idxs: List[int], cam_projs: List[np.ndarray],
                                skel_kps_idxs: List[int], init_param: PoseShapeParam, n_max_iter=5) -> PoseShapeParam:
    """inverse_kinematics.py:241-277: root, angles and the side bone lengths against the 2-D observations."""
    x, _ = _stage_reproj(skel, obs_pose_2d, obs_kps_idxs, cam_projs, skel_kps_idxs, init_param, n_max_iter, 2)
    return PoseShapeParam(x[:3].copy(), x[3:57].reshape(18, 3).copy(), x[57:].copy())


def _stage_3d(skel, obs_pose_3d, obs_kps_idxs, skel_kps_idxs, init_param, n_max_iter, stage):
    _check_idxs(obs_kps_idxs, skel_kps_idxs)
    tgt = np.asarray(obs_pose_3d, np.float64)
    if tgt.shape != (18, 4):
        raise ValueError("obs_pose_3d: expected (18,4) = x, y, z, score per COCO-17 + mid-spine row")
    d = _d()
    x0 = torch.as_tensor(_param_vec(init_param)[None], device=d)
    params, _, info = dev.ik_solve_stages(x0, stage, int(n_max_iter), targets3d=torch.as_tensor(tgt[None].copy(), device=d),
                                          skeleton=skel._to_c())
    return params[0].cpu().numpy(), info[0].cpu().numpy()


def solve_pose(skel: Skeleton, obs_pose_3d: np.ndarray, obs_kps_idxs: List[int], skel_kps_idxs: List[int],
               init_param: PoseShapeParam, n_max_iter=5) -> PoseShapeParam:
    """inverse_kinematics.py:280-307: root and angles against triangulated 3-D joints (weighted by their scores)."""
    x, _ = _stage_3d(skel, obs_pose_3d, obs_kps_idxs, skel_kps_idxs, init_param, n_max_iter, 1)
    return PoseShapeParam(x[:3].copy(), x[3:57].reshape(18, 3).copy(), init_param.bone_lens)


def solve_pose_bone_lens(skel: Skeleton, obs_pose_3d: np.ndarray, obs_kps_idxs: List[int], skel_kps_idxs: List[int],
                         init_param: PoseShapeParam, n_max_iter=5) -> PoseShapeParam:
    """inverse_kinematics.py:310-336: root, angles and side bone lengths against triangulated 3-D joints."""
    x, _ = _stage_3d(skel, obs_pose_3d, obs_kps_idxs, skel_kps_idxs, init_param, n_max_iter, 2)
    return PoseShapeParam(x[:3].copy(), x[3:57].reshape(18, 3).copy(), x[57:].copy())
