"""Synthetic multi-view keypoint generator (the reference has none).

Cameras on a ring of radius 4 m at height 1.5-2.5 m looking at the scene centre, Shelf-like
intrinsics (f = 1080 px, 1032 x 776); P people with roots uniform in a 4 x 4 m area, the 18-joint
skeleton of inverse_kinematics.py:120-173 posed with N(0, 0.3 rad) angles that follow a smooth
random walk (0.02 rad, 2 cm per frame); joints projected into every view, OpenPose-25 layout,
N(0, 2 px) pixel noise, scores U(0.5, 1), 5 % of the joints dropped to (0, 0, 0) the way OpenPose
reports misses; the person order is shuffled independently per view and frame so that association
is not trivial.  numpy.random.default_rng(seed) makes every configuration reproducible.
"""
from __future__ import annotations

import numpy as np

from .device import SKEL_PARENTS, SKEL_SIDE_MAP, skeleton_arrays

# skeleton joint of each COCO keypoint (eyes are synthesised between nose and ear)
_COCO_FROM_SKEL = {0: 15, 3: 16, 4: 17, 5: 9, 6: 12, 7: 10, 8: 13, 9: 11, 10: 14, 11: 1, 12: 4, 13: 2, 14: 5,
                   15: 3, 16: 6}
# OpenPose-25 row of each COCO-17 joint (pose_def.py:111-137)
_OP25_OF_COCO = [0, 16, 15, 18, 17, 5, 2, 6, 3, 7, 4, 12, 9, 13, 10, 14, 11]


def make_cameras(n_views: int, rng: np.random.Generator):
    """-> K (C,3,3), Rt (C,3,4), P (C,3,4) float64."""
    K = np.array([[1080.0, 0, 516.0], [0, 1080.0, 388.0], [0, 0, 1.0]])
    Ks, Rts = [], []
    target = np.array([0.0, 0.0, 1.0])
    for c in range(n_views):
        th = 2 * np.pi * (c + rng.uniform(-0.15, 0.15)) / n_views
        pos = np.array([4.0 * np.cos(th), 4.0 * np.sin(th), rng.uniform(1.5, 2.5)])
        fwd = target - pos
        fwd /= np.linalg.norm(fwd)
        right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd])
        Rts.append(np.concatenate([R, (-R @ pos)[:, None]], axis=1))
        Ks.append(K.copy())
    Ks, Rts = np.array(Ks), np.array(Rts)
    return Ks, Rts, Ks @ Rts


def _rot(axis: int, a: np.ndarray) -> np.ndarray:
    c, s = np.cos(a), np.sin(a)
    R = np.zeros(a.shape + (3, 3))
    i, j = [(1, 2), (2, 0), (0, 1)][axis]
    R[..., axis, axis] = 1.0
    R[..., i, i] = c
    R[..., j, j] = c
    R[..., i, j] = -s
    R[..., j, i] = s
    return R


def batched_fk(root: np.ndarray, euler: np.ndarray, side_lens: np.ndarray) -> np.ndarray:
    """Vectorised FK (R = Rx Ry Rz, parents precede children): root (...,3), euler (...,18,3) -> (...,18,3)."""
    dirs, _ = skeleton_arrays()
    off = dirs * side_lens[..., SKEL_SIDE_MAP, None] if side_lens.ndim > 1 else dirs * side_lens[SKEL_SIDE_MAP, None]
    R = _rot(0, euler[..., 0]) @ _rot(1, euler[..., 1]) @ _rot(2, euler[..., 2])
    Rg = [None] * 18
    pos = [None] * 18
    Rg[0], pos[0] = R[..., 0, :, :], root
    for j in range(1, 18):
        p = SKEL_PARENTS[j]
        pos[j] = pos[p] + np.einsum('...ij,...j->...i', Rg[p], np.broadcast_to(off[..., j, :], pos[p].shape))
        Rg[j] = Rg[p] @ R[..., j, :, :]
    return np.stack(pos, axis=-2)


def scene_walk(n_frames: int, n_people: int, seed: int, segment: int = 0, kappa: float = 4.5e-3):
    """One smooth, BOUNDED walk of the scene's people, cut into consecutive segments of ``n_frames`` frames:
    -> root (n_frames, P, 3), angles (n_frames, P, 18, 3) of segment ``segment``.  Every person wanders around a home position and a
    home pose (drawn from ``seed``): the increments of generate()'s random walk (2 cm and 0.02 rad per frame) with a pull of ``kappa``
    per frame towards home, started in its stationary state, so that a sequence of any length stays in front of the cameras (a plain
    random walk drifts by 9 m over the 200 k frames of BASELINE config 5).  With kappa = 4.5e-3 the wander has a standard deviation of
    0.21 rad / 0.21 m around a home pose of 0.21 rad: joint angles of 0.30 rad in all -- the distribution generate() draws at its chain
    heads, so the poses the solver sees are those of the chain-restart workload of rounds 1 - 5.  Segment s draws its increments from
    its own stream (seed, 3, s) and starts where segment s - 1 ends, so the segments of one seed (and one segment length) tile ONE
    scene whoever generates them: rank r of a sharded run generates segment r, replaying the cheap increments of the segments before
    it for its start state."""
    from scipy.signal import lfilter
    rng0 = np.random.default_rng([seed, 3])
    home_root = np.concatenate([rng0.uniform(-2, 2, size=(n_people, 2)), rng0.uniform(0.95, 1.1, size=(n_people, 1))], -1)
    home_ang = rng0.normal(0, 0.3 / np.sqrt(2.0), size=(n_people, 18, 3))
    sd = np.concatenate([0.02 * np.array([1, 1, 0.1]), np.full(54, 0.02)])                    # per-frame increments
    y0 = rng0.normal(0, 1.0, size=(n_people, 57)) * sd / np.sqrt(1.0 - (1.0 - kappa) ** 2)    # the stationary state
    a = [1.0, -(1.0 - kappa)]
    zi = ((1.0 - kappa) * y0).reshape(1, n_people * 57)
    y = None
    for s in range(segment + 1):
        rng = np.random.default_rng([seed, 3, s])
        d = (rng.normal(0, 1.0, size=(n_frames, n_people, 57)) * sd).reshape(n_frames, n_people * 57)
        y, zi = lfilter([1.0], a, d, axis=0, zi=zi)      # x_t = (1 - kappa) x_{t-1} + d_t, carried across the segments
    y = y.reshape(n_frames, n_people, 57)
    return home_root[None] + y[..., :3], home_ang[None] + y[..., 3:].reshape(n_frames, n_people, 18, 3)


def generate(n_frames: int, n_views: int, n_people: int, seed: int, chain_len: int = 0, dtype=np.float32,
             drop=0.05, pix_sigma=2.0, shuffle=True, frame_seed=None, occlusion=0.0, spurious=0.0, walk=None, segment: int = 0):
    """-> dict(kps25 (F,C,P,25,3), counts (F,C) int32, K, Rt, P, gt_joints (F,P,18,3), gt_order (F,C,P)).

    occlusion: probability that a person is missed entirely by a view in a frame (the view's list gets shorter: ragged counts,
    tracklets seen by one view or none, deaths and re-births); spurious: probability that such a freed slot holds a false
    detection instead (a random pose of plausible size that matches nobody) -- CONDITIONAL on the slot being freed, so a view's
    list is shorter with probability occlusion * (1 - spurious) per person and spurious = 1 turns every occlusion into a ghost
    (round 2 divided by ``occlusion`` here, which made every freed slot a ghost whenever spurious >= occlusion: no ragged counts).
    gt_order is -1 for slots that hold no real person.
    Both default to 0, which leaves the output of earlier versions bit for bit unchanged.

    chain_len > 0 restarts the random walk every chain_len frames (independent sub-sequences).
    Cameras depend on ``seed`` only; ``frame_seed`` (default: seed) drives people, noise and shuffles,
    so ranks can share one calibration and still own different frame shards.

    walk="scene": ONE scene for any number of shards -- the people follow scene_walk(seed) (a smooth bounded walk, no restarts; chain_len
    is ignored), bone lengths come from ``seed``, and this call returns segment ``segment`` of it (frames [segment n_frames,
    (segment + 1) n_frames)); noise, scores, drops and shuffles of a segment come from (seed, 1, segment).  The same seed on every rank
    with segment = rank cuts one sequence into contiguous shards: what the tracklet stitch has identities to carry across.
    walk=None (default): the random walk above, bit for bit the output of earlier versions."""
    K, Rt, Pm = make_cameras(n_views, np.random.default_rng(seed))
    F, C, Pn = n_frames, n_views, n_people
    _, side = skeleton_arrays()
    if walk == "scene":
        root, ang = scene_walk(F, Pn, seed, segment)
        lens = side * np.random.default_rng([seed, 4]).uniform(0.9, 1.1, size=(Pn, 1)) * np.ones((Pn, 11))
        frame_seed = [seed, 5, int(segment)]
        rng = np.random.default_rng(frame_seed + [1])
    elif walk is not None:
        raise ValueError("generate: walk is None (random walk, restarted every chain_len frames) or 'scene'")
    else:
        rng = np.random.default_rng([seed if frame_seed is None else frame_seed, 1])
        L = chain_len if chain_len > 0 else F
        n_chain = (F + L - 1) // L
        root0 = np.concatenate([rng.uniform(-2, 2, size=(n_chain, Pn, 2)), rng.uniform(0.95, 1.1, size=(n_chain, Pn, 1))], -1)
        ang0 = rng.normal(0, 0.3, size=(n_chain, Pn, 18, 3))
        d_root = rng.normal(0, 0.02, size=(n_chain, L, Pn, 3)) * np.array([1, 1, 0.1])
        d_ang = rng.normal(0, 0.02, size=(n_chain, L, Pn, 18, 3))
        d_root[:, 0] = 0
        d_ang[:, 0] = 0
        root = (root0[:, None] + np.cumsum(d_root, axis=1)).reshape(n_chain * L, Pn, 3)[:F]
        ang = (ang0[:, None] + np.cumsum(d_ang, axis=1)).reshape(n_chain * L, Pn, 18, 3)[:F]
        lens = side * rng.uniform(0.9, 1.1, size=(Pn, 1)) * np.ones((Pn, 11))
    joints = batched_fk(root, ang, np.broadcast_to(lens, (F, Pn, 11)))  # (F,P,18,3)

    # 3-D points of the OpenPose-25 layout
    X = np.zeros((F, Pn, 25, 3))
    have = np.zeros(25, dtype=bool)
    for coco, sk in _COCO_FROM_SKEL.items():
        X[:, :, _OP25_OF_COCO[coco]] = joints[:, :, sk]
        have[_OP25_OF_COCO[coco]] = True
    nose, lear, rear = joints[:, :, 15], joints[:, :, 16], joints[:, :, 17]
    X[:, :, 16] = 0.6 * nose + 0.4 * lear  # L_Eye
    X[:, :, 15] = 0.6 * nose + 0.4 * rear  # R_Eye
    X[:, :, 1] = joints[:, :, 8]           # Neck
    X[:, :, 8] = joints[:, :, 0]           # Mid_Hip
    for op, ankle, dx in ((19, 3, 0.12), (20, 3, 0.10), (21, 3, -0.05), (22, 6, 0.12), (23, 6, 0.10), (24, 6, -0.05)):
        X[:, :, op] = joints[:, :, ankle] + np.array([0.0, dx, -0.05])
    # project: (F,C,P,25)
    h = np.einsum('cik,fpjk->fcpji', Pm[:, :, :3], X) + Pm[None, :, None, None, :, 3]
    uv = h[..., :2] / h[..., 2:3]
    uv = uv + rng.normal(0, pix_sigma, size=uv.shape)
    score = rng.uniform(0.5, 1.0, size=uv.shape[:-1])
    miss = rng.uniform(size=score.shape) < drop
    kps = np.concatenate([uv, score[..., None]], axis=-1)
    kps[miss] = 0.0
    order = np.broadcast_to(np.arange(Pn), (F, C, Pn)).copy()
    if shuffle:
        order = rng.permuted(order, axis=-1)
        kps = np.take_along_axis(kps, order[..., None, None], axis=2)
    counts = np.full((F, C), Pn, dtype=np.int32)
    if occlusion > 0.0 or spurious > 0.0:
        rng2 = np.random.default_rng((frame_seed + [2]) if isinstance(frame_seed, list) else [seed if frame_seed is None else frame_seed, 2])
        gone = rng2.uniform(size=(F, C, Pn)) < occlusion
        ghost = gone & (rng2.uniform(size=(F, C, Pn)) < spurious)
        centre = rng2.uniform([100.0, 100.0], [900.0, 650.0], size=(F, C, Pn, 1, 2))
        fake = np.concatenate([centre + rng2.normal(0, 60.0, size=(F, C, Pn, 25, 2)), rng2.uniform(0.3, 0.9, size=(F, C, Pn, 25, 1))], axis=-1)
        kps = np.where(ghost[..., None, None], fake, kps)
        order = np.where(gone, -1, order)
        keep = ~gone | ghost
        # compact every view's list (kept slots first, their order preserved), zero the rest
        idx = np.argsort(~keep, axis=-1, kind="stable")
        kps = np.take_along_axis(kps, idx[..., None, None], axis=2)
        order = np.take_along_axis(order, idx, axis=2)
        counts = keep.sum(axis=-1).astype(np.int32)
        kps[np.arange(Pn)[None, None, :] >= counts[..., None]] = 0.0
        order = np.where(np.arange(Pn)[None, None, :] >= counts[..., None], -1, order)
    return dict(kps25=np.ascontiguousarray(kps.astype(dtype)), counts=counts, K=K, Rt=Rt, P=Pm,
                gt_joints=joints, gt_order=order.astype(np.int32))
