"""``ForwardKinematics`` of the reference's ``kinematics.py`` (:11-31) -- named by the north star, dead code in the reference (nothing
imports it; the live FK is ``inverse_kinematics.foward_kinematics`` = ``mvmc_fk``).  Host NumPy like the original, for a skeleton
object with ``offset`` (J, 3), ``topology`` (parent indices) and ``chosen_joints``; rotations are SciPy ``Rotation`` objects.

Two behaviours of the original are part of its results and are kept:
* the chain product also runs for joint 0 with ``topology[0]`` as its parent (:25-27).  With the usual ``topology[0] == -1`` the root's
  global transform is therefore  local[J-1] @ local[0]  (index -1 reads the LAST joint, whose entry is still its local transform at
  that point), and every descendant inherits it;
* the returned positions are divided by their own z (:29-30): rows are (x/z, y/z, 1).
"""
from __future__ import annotations

from typing import Sequence

import numpy as np


class ForwardKinematics:
    def __init__(self, skel):
        self.offset = skel.offset
        self.parents = skel.topology
        self.chosen_joints = skel.chosen_joints
        self.n_joints = len(self.offset)

    def forward(self, rotations: Sequence) -> np.ndarray:
        n = self.n_joints
        local = np.zeros((n, 4, 4))
        local[:, 3, 3] = 1.0
        local[:, :3, :3] = np.stack([rotations[j].as_matrix() for j in range(n)])
        local[1:, :3, 3] = np.asarray(self.offset, dtype=float)[1:]
        glob = local.copy()
        for j in range(n):                       # joint 0 included, see the module text
            glob[j] = glob[self.parents[j]] @ local[j]
        pos = glob[:, :3, 3]
        return pos / pos[:, 2:3]
