"""Batched hot path on one GPU: association -> triangulation -> IK over a shard of frames.

``HotPath`` owns the frame-invariant state (calibration, fundamental matrices, skeleton) and runs
the per-frame stages as a fixed sequence of kernel launches on the current stream, with no host
synchronisation in between (every buffer is sized by the static shape, ragged counts stay on the
device).  Stage order follows MvTracker.update_4d for frames without live tracklets
(motion_capture.py:873-958 -> match_spatial :597-631 -> MvTracklet.__init__ :312-332).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import device as dev


class HotPath:
    def __init__(self, K, Rt, device="cuda:0", k_max: Optional[int] = None, v_max: Optional[int] = None):
        d = torch.device(device)
        self.device = d
        self.K = torch.as_tensor(K, dtype=torch.float64, device=d).contiguous()
        self.Rt = torch.as_tensor(Rt, dtype=torch.float64, device=d).contiguous()
        self.P = torch.bmm(self.K, self.Rt).contiguous()  # load_calib: P = K @ Rt (motion_capture.py:266)
        self.F = dev.fmats(self.K, self.Rt)
        self.skeleton = dev.make_skeleton()
        self.k_max = k_max
        self.v_max = v_max

    # -- stages ----------------------------------------------------------------------------------
    def associate(self, kps: torch.Tensor, counts: Optional[torch.Tensor] = None, want_mats=False):
        """IN-1/2 + AS-1..6 for every frame of the batch (frames are independent here)."""
        kps17, cnt = dev.ingest(kps, counts)
        P = kps17.shape[2]
        D, S = dev.affinity(kps17, cnt, self.F, want_D=want_mats)
        res = dev.als_associate(S, cnt, g_max=P, want_mats=want_mats)
        res.update(kps17=kps17, counts=cnt, D=D, S=S)
        return res

    def triangulate(self, assoc: dict):
        """TR-1/2 for every cluster: members (F,K,V), n_members (F,K), pts3d (F,K,17,4)."""
        kps17, cnt = assoc["kps17"], assoc["counts"]
        C, P = kps17.shape[1], kps17.shape[2]
        # defaults that the frame's size rules out exceeding (the reference has no caps: motion_capture.py:417-446): a cluster needs two
        # members, so at most C P / 2 of them; one holds at most C P poses
        k_max = self.k_max or max(1, (C * P) // 2)
        v_max = self.v_max or min(C * P, 64)
        mem, nm = dev.cluster_members(assoc["labels"], cnt, P, k_max, v_max)
        pts = dev.dlt(kps17, self.P, mem)
        return dict(members=mem, n_members=nm, pts3d=pts)

    def solve_cold(self, assoc: dict, tri: dict, max_nfev=50):
        """IK-1..4 cold start for every cluster with >= 2 views (MvTracklet.__init__)."""
        mem = tri["members"]
        F, Kc, V = mem.shape
        params, joints, info = dev.ik_solve(assoc["kps17"], self.P, mem.reshape(F * Kc, V), None, None,
                                            max_nfev_cold=max_nfev, skeleton=self.skeleton)
        return dict(params=params.reshape(F, Kc, 68), joints=joints.reshape(F, Kc, 18, 3),
                    info=info.reshape(F, Kc, 8))

    def run(self, kps: torch.Tensor, counts: Optional[torch.Tensor] = None, with_ik=True, max_nfev_cold=50):
        assoc = self.associate(kps, counts)
        tri = self.triangulate(assoc)
        out = dict(labels=assoc["labels"], n_clusters=assoc["n_clusters"], als_iters=assoc["iters"], **tri)
        if with_ik:
            out.update(self.solve_cold(assoc, tri, max_nfev_cold))
        return out
