"""CPU oracle of the temporal layer: match_spatial_time (AS-7) and the tracker state machine
(TK-1), restated in NumPy on top of oracle_np.py.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE (see oracle/oracle_np.py header).

Follows motion_capture.py:634-826 (match_spatial_time), :829-835 (associate_tracking),
:312-400 (MvTracklet) and :873-963 (MvTracker.update_4d).  Pinned by
tests/golden/shelf_tracker.npz (the reference's own run over the Shelf sequence).
"""
from __future__ import annotations

import numpy as np

import oracle_np as o

TENTATIVE, CONFIRMED, DEAD = 1, 2, 3


class Tracklet:
    def __init__(self, tid, frame, param, joints):
        self.tid, self.state, self.hits, self.length = tid, TENTATIVE, 1, 1
        self.param, self.joints = param, joints
        self.time_since_update = 0
        self.frames = [frame]


def associate(tracklets, views_kps, Ps, Ks=None, Rts=None):
    """-> (tracklet_matches {tracklet index: [(view, local pose idx)]}, new_matches [[(view, idx)]])
    views_kps: per view list of (17,3) poses (already filtered)."""
    tm, nm = {}, []
    if tracklets:
        D, dim = o.spatial_time_distance([t.joints for t in tracklets], views_kps, Ps)
        _, S = o.spatial_time_affinity(D)
        mm, _ = o.match_als(S, dim)
        T = len(tracklets)
        for cl in o.parse_match_result(mm, len(mm), dim):
            tidx = next((gi for _, _, gi in cl if gi < T), -1)
            match = []
            for grp, loc, gi in cl:
                if gi >= T and (grp - 1) not in [v for v, _ in match]:
                    match.append((grp - 1, loc))
            if tidx >= 0:
                if match:
                    tm[tidx] = match
            elif match:
                nm.append(match)
    else:
        # match_spatial: views with zero people are skipped by the group logic
        pts = [p[:, :2] for v in views_kps for p in v]
        dim = np.concatenate([[0], np.cumsum([len(v) for v in views_kps])]).tolist()
        F = o.pairwise_f_mats(Ks, Rts)
        _, S = o.geometry_affinity(np.array(pts), F, dim)
        mm, _ = o.match_als(S, dim)
        for cl in o.parse_match_result(mm, len(mm), dim):
            nm.append([(grp, loc) for grp, loc, _ in cl])  # no per-view dedupe on this path (:621-626)
    return tm, nm


class OracleTracker:
    """MvTracker (motion_capture.py:840-963) with n_inits = 3, max_age = 0."""

    def __init__(self, Ks, Rts, Ps, solver=None):
        self.Ks, self.Rts, self.Ps = Ks, Rts, Ps
        self.tracklets, self.n_dead, self.next_id = [], 0, 0
        self.solver = solver or (lambda poses, projs, init: o.pose_solver_solve(poses, projs, init))
        self.solves = []

    def update(self, frame_idx, views_kps):
        for t in self.tracklets:
            t.time_since_update += 1
        tm, nm = associate(self.tracklets, views_kps, self.Ps, self.Ks, self.Rts)
        for ti, t in enumerate(self.tracklets):
            if ti in tm:
                m = tm[ti]
                if len(m) >= 2:
                    poses = [views_kps[v][l] for v, l in m]
                    param, joints = self.solver(poses, [self.Ps[v] for v, _ in m], t.param)
                    self.solves.append((frame_idx, False, len(m), param, joints))
                    t.param, t.joints = param, joints
                    t.frames.append(frame_idx)
                    t.length += 1
                    t.time_since_update = 0
                    t.hits += 1
                    if t.state == TENTATIVE and t.hits >= 3:
                        t.state = CONFIRMED
            else:
                t.state = DEAD  # Tentative dies at once; Confirmed dies because time_since_update > max_age = 0
        for m in nm:
            if len(m) >= 2:
                poses = [views_kps[v][l] for v, l in m]
                param, joints = self.solver(poses, [self.Ps[v] for v, _ in m], None)
                self.solves.append((frame_idx, True, len(m), param, joints))
                self.tracklets.append(Tracklet(self.next_id, frame_idx, param, joints))
                self.next_id += 1
        self.n_dead += sum(t.state == DEAD for t in self.tracklets)
        self.tracklets = [t for t in self.tracklets if t.state != DEAD]
