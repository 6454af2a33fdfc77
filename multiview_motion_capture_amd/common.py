"""Calib / FrameData containers (mirror of the reference's common.py:7-25)."""
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np

from .pose_def import Pose


@dataclass
class Calib:
    K: np.ndarray       # 3x3
    Rt: np.ndarray      # 3x4
    P: np.ndarray       # 3x4
    Kr_inv: np.ndarray  # 3x3
    img_wh_size: Tuple[int, int]

    @property
    def cam_loc(self):
        return -self.Rt[:3, :3].T @ self.Rt[:3, 3]

    @classmethod
    def from_k_rt(cls, K, Rt, img_wh_size=(0, 0)):
        """load_calib's arithmetic (motion_capture.py:262-270): P = K Rt, Kr_inv = R^T K^-1."""
        K = np.asarray(K, np.float64).reshape(3, 3)
        Rt = np.asarray(Rt, np.float64).reshape(3, 4)
        return cls(K=K, Rt=Rt, P=K @ Rt, Kr_inv=Rt[:3, :3].T @ np.linalg.inv(K), img_wh_size=tuple(img_wh_size))


@dataclass
class FrameData:
    frame_idx: int
    poses: Dict[int, Pose]
    calib: Calib
    view_id: int
