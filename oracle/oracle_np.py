"""CPU oracle: NumPy restatement of the reference's per-frame hot path.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the package ``multiview_motion_capture_amd`` never does.

Every function restates one row of SURVEY.md section 8(a) and cites the
reference file:line (relative to the reference checkout) it follows.  The
restatement is pinned by the golden vectors in ``tests/golden`` which
``oracle/gen_golden.py`` produced by importing and *running the reference
itself* in the build container (tests/test_oracle_golden.py).

Parity status:
  * pinned against outputs of the reference run here: F-matrices, affinity,
    ALS, closure, cluster parsing, DLT (+post-optimise), FK (also against the
    725 known-answer poses shipped in data/shelf/tracklets/traclets.pkl),
    IK (cold + warm), epipolar / reprojection error, spatio-temporal matching.
  * "parity unpinned": ``epilines`` restates OpenCV's
    ``computeCorrespondEpilines`` (calib3d/fundam.cpp).  OpenCV is a
    third-party dependency of the reference that is neither vendored nor
    installed; no reference test pins that call.
  * third-party numerics that ARE installed and are used as-is, exactly as the
    reference uses them: ``numpy.linalg.svd/inv/det`` and
    ``scipy.optimize.least_squares`` (reference pins scipy 1.3.2 / numpy
    1.17.4 in requirements.txt:33,55; the container has 1.15.3 / 2.2.6).
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import least_squares

# ----------------------------------------------------------------------------
# schema constants (pose_def.py)
# ----------------------------------------------------------------------------
# OpenPose-25 row of each COCO-17 joint: pose_def.py:72-96 (COCO order),
# :111-137 (OpenPose order), gather at :262-270.
OPENPOSE25_TO_COCO17 = np.array([0, 16, 15, 18, 17, 5, 2, 6, 3, 7, 4, 12, 9, 13, 10, 14, 11])

N_SKEL = 18  # BASIC_18, pose_def.py:202-219
# parents: pose_def.py:181-200,222-224
SKEL_PARENTS = np.array([-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 9, 10, 8, 12, 13, 8, 15, 15])
# COCO indices (pose_def.py:72-96)
COCO_L_SHOULDER, COCO_R_SHOULDER, COCO_L_HIP, COCO_R_HIP = 5, 6, 11, 12

# skeleton joint <-> observed keypoint pairs used by the IK residual
# (inverse_kinematics.py:366-368 via pose_def.get_common_kps_idxs_1, with the
# synthetic mid-spine appended as observed keypoint 17, :370-378)
IK_SKEL_IDX = np.array([1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15, 16, 17])
IK_OBS_IDX = np.array([11, 13, 15, 12, 14, 16, 17, 5, 7, 9, 6, 8, 10, 0, 3, 4])
# common joints BASIC_18 -> COCO used by reprojection_error
# (motion_capture.py:404 via pose_def.get_common_kps_idxs :278-288)
REPROJ_SKEL_IDX = np.array([1, 2, 3, 4, 5, 6, 9, 10, 11, 12, 13, 14, 15, 16, 17])
REPROJ_COCO_IDX = np.array([11, 13, 15, 12, 14, 16, 5, 7, 9, 6, 8, 10, 0, 3, 4])

# skeleton constants: inverse_kinematics.py:121-140 (offsets), :22-26
# (direction / length split), :150-163 (left+mid side lengths and the map)
_SKEL_OFFSETS = np.array([
    [0, 0, 0], [0.15, 0, 0], [0, 0, -0.5], [0, 0, -0.5], [-0.15, 0, 0], [0, 0, -0.5],
    [0, 0, -0.5], [0, 0, 0.3], [0, 0, 0.3], [0.2, 0, 0], [0.3, 0, 0], [0.3, 0, 0],
    [-0.2, 0, 0], [-0.3, 0, 0], [-0.3, 0, 0], [0, -0.02, 0.15], [0.07, 0.02, 0.1],
    [-0.07, 0.02, 0.1]], dtype=np.float64)
SIDE_TO_FULL = np.array([7, 0, 1, 2, 0, 1, 2, 8, 9, 3, 4, 5, 3, 4, 5, 10, 6, 6])
# skeleton joints whose lengths form the 11 side lengths (7 left + 4 mid)
_SIDE_JOINTS = np.array([1, 2, 3, 9, 10, 11, 16, 0, 7, 8, 15])


def skeleton_constants():
    """(ref_bone_dirs[18,3], ref_side_bone_lens[11]) -- inverse_kinematics.py:120-173."""
    lens = np.linalg.norm(_SKEL_OFFSETS, axis=-1)
    dirs = _SKEL_OFFSETS.copy()
    dirs[1:] = dirs[1:] / lens[1:, None]
    return dirs, lens[_SIDE_JOINTS].copy()


# ----------------------------------------------------------------------------
# IN-1 / IN-2 / IN-3
# ----------------------------------------------------------------------------
def openpose25_to_coco17(kps25):
    """(...,25,3) -> (...,17,3).  pose_def.py:262-270, motion_capture.py:980-983."""
    return np.asarray(kps25)[..., OPENPOSE25_TO_COCO17, :]


def pose_is_good(kps17, min_score=0.01, n_min_valid=4, min_bb=5):
    """filter_bad_pose, motion_capture.py:1023-1043.  kps17 (17,3) -> bool keep."""
    valid = kps17[:, 2] > min_score
    if valid.sum() < n_min_valid:
        return False
    xy = kps17[valid, :2]
    size = xy.max(axis=0) - xy.min(axis=0)
    return not bool(np.any(size < min_bb))


def calib_from_k_rt(K, Rt):
    """load_calib (json branch), motion_capture.py:262-270 -> (P, Kr_inv)."""
    K = np.asarray(K, np.float64).reshape(3, 3)
    Rt = np.asarray(Rt, np.float64).reshape(3, 4)
    return K @ Rt, Rt[:, :3].T @ np.linalg.inv(K)


# ----------------------------------------------------------------------------
# AS-1  pairwise fundamental matrices
# ----------------------------------------------------------------------------
def _skew(x):
    return np.array([[0, -x[2], x[1]], [x[2], 0, -x[0]], [-x[1], x[0], 0]], dtype=np.float64)


def pairwise_f_mats(Ks, Rts):
    """calc_pairwise_f_mats, mv_math_util.py:267-285.  f64 math, f32 storage."""
    C = len(Ks)
    F = np.zeros((C, C, 3, 3), dtype=np.float32)
    for i in range(C):
        K0, R0, T0 = Ks[i], Rts[i][:, :3], Rts[i][:, 3]
        for j in range(C):
            K1, R1, T1 = Ks[j], Rts[j][:, :3], Rts[j][:, 3]
            R01 = R0 @ R1.T
            f = np.linalg.inv(K0).T @ R01 @ K1.T @ _skew(K1 @ R1 @ R0.T @ (T0 - R01 @ T1))
            F[i, j] += f.astype(np.float32)
            if F[i, j].sum() == 0:
                F[i, j] += np.float32(1e-12)
    return F


# ----------------------------------------------------------------------------
# AS-2 / AS-3  epipolar distance + affinity
# ----------------------------------------------------------------------------
def epilines(F, pts, which_image):
    """OpenCV computeCorrespondEpilines restated (see module docstring).
    pts (N,2) -> (N,3) with a^2+b^2 = 1."""
    Fm = np.asarray(F, np.float64)
    if which_image == 2:
        Fm = Fm.T
    h = np.concatenate([pts, np.ones((len(pts), 1))], axis=1)
    l = h @ Fm.T
    nu = l[:, 0] ** 2 + l[:, 1] ** 2
    sc = np.ones_like(nu)
    nz = nu != 0
    sc[nz] = 1.0 / np.sqrt(nu[nz])
    return l * sc[:, None]


def projected_distance(pts0, pts1, F):
    """mv_math_util.py:288-317: mean_j |line(F^T x0_j) . x1_j| -> (P0,P1)."""
    nj = pts0.shape[1]
    lines = epilines(F, pts0.reshape(-1, 2), 2).reshape(-1, nj, 3)
    h1 = np.concatenate([pts1, np.ones(pts1.shape[:2] + (1,))], axis=2)
    d = np.abs(np.einsum('ajk,bjk->abj', lines, h1))
    return d.mean(axis=2)


def np_pairwise_sum_f32(a):
    """NumPy's float32 pairwise summation of a contiguous 1-D array
    (numpy/_core/src/umath/loops_utils.h.src, PW_BLOCKSIZE=128), restated
    explicitly because the C / HIP code has to reproduce the f32 statistics of
    geometry_affinity (mv_math_util.py:348) bit for bit."""
    a = np.asarray(a, np.float32)
    n = len(a)
    f = np.float32
    if n < 8:
        res = f(-0.0)  # numpy seeds the short loop with the reduction identity
        res = f(0.0) if n == 0 else res
        for x in a:
            res = f(res + x)
        return res
    if n <= 128:
        r = [f(a[k]) for k in range(8)]
        i = 8
        while i < n - (n % 8):
            for k in range(8):
                r[k] = f(r[k] + a[i + k])
            i += 8
        res = f(f(f(r[0] + r[1]) + f(r[2] + r[3])) + f(f(r[4] + r[5]) + f(r[6] + r[7])))
        while i < n:
            res = f(res + a[i])
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return f(np_pairwise_sum_f32(a[:n2]) + np_pairwise_sum_f32(a[n2:]))


def affinity_from_distance(D):
    """mv_math_util.py:348-350, all float32."""
    D = np.asarray(D, np.float32)
    A = -(D - D.mean()) / D.std()
    return 1 / (1 + np.exp(-5 * A))


def geometry_affinity(points_set, Fs, dim_group):
    """mv_math_util.py:320-351 -> (D f32 (M,M), S f32 (M,M))."""
    M = points_set.shape[0]
    D = np.full((M, M), 50, dtype=np.float32)
    np.fill_diagonal(D, 0)
    ng = len(dim_group) - 1
    for a in range(ng):
        for b in range(a + 1, ng):
            a0, a1, b0, b1 = dim_group[a], dim_group[a + 1], dim_group[b], dim_group[b + 1]
            if a0 == a1 or b0 == b1:
                continue
            pa, pb = points_set[a0:a1], points_set[b0:b1]
            blk = 0.5 * (projected_distance(pa, pb, Fs[a, b]) + projected_distance(pb, pa, Fs[b, a]).T)
            D[a0:a1, b0:b1] = blk
            D[b0:b1, a0:a1] = D[a0:a1, b0:b1].T
    return D, affinity_from_distance(D)


# ----------------------------------------------------------------------------
# AS-4 / AS-5 / AS-6  ALS matching, closure, cluster parsing
# ----------------------------------------------------------------------------
ALS_ALPHA, ALS_BETA, ALS_MU0, ALS_TOL, ALS_MAXITER = 50, 0.1, 64, 1e-4, 1000


def transform_closure(x_bin):
    """mv_association.py:99-121.  Only k = N-1 survives the overwrite (:105-110)."""
    xb = np.asarray(x_bin).astype(bool)
    n = xb.shape[0]
    k = n - 1
    temp = xb | (xb[:, k][:, None] & xb[k, :][None, :])
    out = np.zeros((n, n), dtype=xb.dtype)
    vis = np.zeros(n, dtype=bool)
    for i in range(n):
        if vis[i]:
            continue
        members = np.nonzero(temp[i])[0]
        vis[members] = True
        out[members, i] = True
    return out


def match_als(W, dim_group, return_iters=False):
    """mv_association.py:222-318.  Dtype propagation follows NumPy exactly:
    a float32 W (from geometry_affinity) keeps iteration 1's X update in f32."""
    dim_group = list(dim_group)
    n = W.shape[0]
    rank = min(n, int(max(np.diff(dim_group))) * 2)
    W = 0.5 * (W + W.T)
    X = W.copy()
    Z = W.copy()
    Y = np.zeros_like(W)
    mu = ALS_MU0
    A = np.random.RandomState(0).rand(n, rank)
    eye = np.eye(rank)
    iters = ALS_MAXITER
    for it in range(ALS_MAXITER):
        X0 = X
        X = Z - (Y - W + ALS_BETA) / mu
        B = (np.linalg.inv(A.T @ A + ALS_ALPHA / mu * eye) @ (A.T @ X)).T
        A = (np.linalg.inv(B.T @ B + ALS_ALPHA / mu * eye) @ (B.T @ X.T)).T
        X = A @ B.T
        Z = X + Y / mu
        for g in range(len(dim_group) - 1):
            Z[dim_group[g]:dim_group[g + 1], dim_group[g]:dim_group[g + 1]] = 0
        Z[np.arange(n), np.arange(n)] = 1
        Z[Z < 0] = 0
        Z[Z > 1] = 1
        Y = Y + mu * (X - Z)
        p_res = np.linalg.norm(X - Z) / n
        d_res = mu * np.linalg.norm(X - X0) / n
        if p_res < ALS_TOL and d_res < ALS_TOL:
            iters = it + 1
            break
        if p_res > 10 * d_res:
            mu = 2 * mu
        elif d_res > 10 * p_res:
            mu = mu / 2
    X = 0.5 * (X + X.T)
    x_bin = X > 0.5
    match_mat = transform_closure(x_bin)
    if return_iters:
        return match_mat, x_bin, iters
    return match_mat, x_bin


# ---- match_svt (mv_association.py:321-411) with its doubly-stochastic projection (:15-60) ----
def _proj2pav(y):
    """mv_association.py:49-60: clip negatives; a vector summing below 1 is kept, otherwise projected on the simplex."""
    y = np.where(y < 0, 0, y).astype(y.dtype)
    if y.sum(dtype=y.dtype) < 1:
        return y
    u = np.sort(y)[::-1]
    sv = np.cumsum(u, dtype=y.dtype)
    to_find = u > (sv - 1) / np.arange(1, len(u) + 1, dtype=y.dtype)
    rho = np.nonzero(to_find)[0][-1]
    theta = max(y.dtype.type(0), (sv[rho] - 1) / y.dtype.type(rho + 1))
    return np.maximum(y - theta, 0).astype(y.dtype)


def _proj2dpam(Y, tol):
    """mv_association.py:15-32: alternating row / column projections with Dykstra corrections, at most 10 rounds."""
    X0, X, I2 = Y, Y, 0
    for _ in range(10):
        T = X0 + I2
        X1 = np.stack([_proj2pav(T[i]) for i in range(T.shape[0])])
        I1 = X1 - T
        T = X0 + I1
        X2 = np.stack([_proj2pav(T[:, j]) for j in range(T.shape[1])], axis=1)
        I2 = X2 - T
        chg = np.abs(X2 - X).sum(dtype=Y.dtype) / X.size
        X = X2
        if chg < tol:
            return X
    return X


def match_svt(S, dim_group, alpha=0.1, lam=50, mu=64, tol=5e-4, max_iter=20, dual_stochastic=True, return_info=False):
    """mv_association.py:321-411 (pSelect = 1).  The dtype of S carries through, as with torch.from_numpy: the float32 affinity of
    geometry_affinity keeps the whole iteration (SVD included) in float32."""
    S = np.array(S)   # the reference writes the diagonal of its argument; the oracle works on a copy
    dt = S.dtype.type
    N = S.shape[0]
    dim_group = [int(v) for v in dim_group]
    S[np.arange(N), np.arange(N)] = 0
    S = (S + S.T) / dt(2)
    X = S.copy()
    Y = np.zeros_like(S)
    W = dt(alpha) - S
    mu = float(mu)
    n_iter = max_iter
    for it in range(max_iter):
        X0 = X
        U, s, Vt = np.linalg.svd((dt(1.0 / mu) * Y + X).astype(S.dtype))
        ds = np.maximum(s - dt(lam / mu), 0).astype(S.dtype)
        Q = ((U * ds) @ Vt).astype(S.dtype)
        X = (Q - (W + Y) / dt(mu)).astype(S.dtype)
        for g in range(len(dim_group) - 1):
            X[dim_group[g]:dim_group[g + 1], dim_group[g]:dim_group[g + 1]] = 0
        X[np.arange(N), np.arange(N)] = 1
        X[X < 0] = 0
        X[X > 1] = 1
        if dual_stochastic:
            for gi in range(len(dim_group) - 1):
                r0, r1 = dim_group[gi], dim_group[gi + 1]
                for gj in range(len(dim_group) - 1):
                    c0, c1 = dim_group[gj], dim_group[gj + 1]
                    if r1 > r0 and c1 > c0:
                        X[r0:r1, c0:c1] = _proj2dpam(X[r0:r1, c0:c1].copy(), dt(1e-2))
        X = ((X + X.T) / dt(2)).astype(S.dtype)
        Y = (Y + dt(mu) * (X - Q)).astype(S.dtype)
        p_res = float(np.linalg.norm(X - Q)) / N
        d_res = mu * float(np.linalg.norm(X - X0)) / N
        if p_res < tol and d_res < tol:
            n_iter = it
            break
        if p_res > 10 * d_res:
            mu = 2 * mu
        elif d_res > 10 * p_res:
            mu = mu / 2
    X = (X + X.T) / dt(2)
    x_bin = X > 0.5
    match_mat = transform_closure(x_bin)
    if return_info:
        return match_mat, x_bin, dict(iter=n_iter, X=X)
    return match_mat, x_bin


def parse_match_result(match_mat, n, dim_group):
    """motion_capture.py:417-446 -> clusters of (group, local, global)."""
    mm = np.asarray(match_mat).astype(np.float64)
    keep = np.nonzero(mm.sum(axis=0) > 1.9)[0]
    if keep.size == 0:
        # the reference's torch reshape(n, -1) of an empty tensor raises here
        raise RuntimeError("parse_match_result: no cluster with >= 2 members")
    b = mm[:, keep] > 0.9
    clusters = [[] for _ in range(len(keep))]
    for row in range(n):
        if b[row].any():
            clusters[int(np.argmax(b[row]))].append(row)
    dg = np.asarray(dim_group)
    out = []
    for members in clusters:
        cur = []
        for idx in members:
            grp = int(np.nonzero(dg <= idx)[0][-1])
            cur.append((grp, int(idx - dg[grp]), int(idx)))
        if cur:
            out.append(cur)
    return out


def cluster_labels(match_mat, n):
    """Flat form of parse_match_result used by the batched device path:
    label[i] = cluster ordinal of node i (ascending representative), -1 if the
    node is in no cluster of >= 2 members.  Same rule, motion_capture.py:419-425."""
    mm = np.asarray(match_mat).astype(np.float64)
    keep = np.nonzero(mm.sum(axis=0) > 1.9)[0]
    lab = -np.ones(n, dtype=np.int32)
    if keep.size == 0:
        return lab
    b = mm[:, keep] > 0.9
    for row in range(n):
        if b[row].any():
            lab[row] = int(np.argmax(b[row]))
    return lab


# ----------------------------------------------------------------------------
# AS-8 / AS-9  epipolar error between two 2-D poses, reprojection error
# ----------------------------------------------------------------------------
def fundamental_from_projections(P1, P2):
    """get_fundamental_matrix, mv_math_util.py:57-77: F[i,j] = det([X_j; Y_i])."""
    X = [P1[[1, 2]], P1[[2, 0]], P1[[0, 1]]]
    Y = [P2[[1, 2]], P2[[2, 0]], P2[[0, 1]]]
    F = np.zeros((3, 3), dtype=P1.dtype)
    for i in range(3):
        for j in range(3):
            F[i, j] = np.linalg.det(np.vstack([X[j], Y[i]]))
    return F


def epipolar_error(P1, kps1, sc1, P2, kps2, sc2, min_score=0.05, invalid=np.nan):
    """calc_epipolar_error, mv_math_util.py:80-115."""
    if len(kps1) == 0:
        return invalid
    F = fundamental_from_projections(P1, P2)
    l12 = epilines(F, kps1, 1)
    l21 = epilines(F, kps2, 2)
    valid = (np.asarray(sc1).ravel() * np.asarray(sc2).ravel()) > min_score
    if not valid.any():
        return invalid
    total, cnt = 0, 0
    for j in np.nonzero(valid)[0]:
        a, b, c = l12[j]
        d1 = abs(a * kps2[j, 0] + b * kps2[j, 1] + c) / np.sqrt(a ** 2 + b ** 2)
        a, b, c = l21[j]
        d2 = abs(a * kps1[j, 0] + b * kps1[j, 1] + c) / np.sqrt(a ** 2 + b ** 2)
        total = total + 0.5 * (d1 + d2)
        cnt += 1
    return total / cnt


def reprojection_error(joints3d, kps2d, sc2d, P, min_score=0.05, invalid=np.nan):
    """motion_capture.py:403-414.  joints3d (18,3) BASIC_18 (score == 1),
    kps2d (17,2) COCO, sc2d (17,)."""
    X = joints3d[REPROJ_SKEL_IDX]
    h = P @ np.concatenate([X, np.ones((len(X), 1))], axis=1).T
    uv = (h[:2] / (1e-5 + h[2])).T
    s2 = np.asarray(sc2d).ravel()[REPROJ_COCO_IDX]
    mask = (s2 * 1.0) > min_score
    if not mask.any():
        return invalid
    e = np.linalg.norm(uv[mask] - kps2d[REPROJ_COCO_IDX][mask], axis=-1)
    return np.mean(e)


def spatial_time_affinity(D):
    """motion_capture.py:744-756: NaN fill, fixed-statistics sigmoid, clamps."""
    D = np.array(D, dtype=np.float64)
    mx = np.nanmax(D)
    D[np.isnan(D)] = mx + 1.0
    S = 1 / (1 + np.exp(5 * ((D - 15) / 30)))
    S[S < 1e-3] = 0
    S[S > 1.0] = 1.0
    return D, S


def spatial_time_distance(track_joints, views_kps, Ps):
    """Distance matrix of match_spatial_time, motion_capture.py:651-740.
    track_joints: list of (18,3); views_kps: per view list of (17,3); Ps: per view (3,4).
    Node order = tracklets, then 2-D poses by view.  Returns (D with NaN, dim_groups)."""
    T = len(track_joints)
    cam_of = [-1] * T
    nodes = [('3d', j) for j in track_joints]
    parts = [0, T]
    for v, poses in enumerate(views_kps):
        for p in poses:
            nodes.append(('2d', p))
            cam_of.append(v)
        parts.append(len(poses))
    dim_groups = np.cumsum(parts).tolist()
    n = len(nodes)
    D = np.zeros((n, n))
    for i in range(n):
        for j in range(n):
            if i == j:
                continue
            if cam_of[i] >= 0 and cam_of[i] == cam_of[j]:
                D[i, j] = np.nan
                continue
            ki, kj = nodes[i][0], nodes[j][0]
            if ki == '2d' and kj == '2d':
                a, b = nodes[i][1], nodes[j][1]
                D[i, j] = epipolar_error(Ps[cam_of[i]], a[:, :2], a[:, 2], Ps[cam_of[j]], b[:, :2], b[:, 2],
                                         0.1, np.nan)
            elif ki == '2d' and kj == '3d':
                a = nodes[i][1]
                D[i, j] = reprojection_error(nodes[j][1], a[:, :2], a[:, 2], Ps[cam_of[i]], 0.1, np.nan)
            elif ki == '3d' and kj == '2d':
                b = nodes[j][1]
                D[i, j] = reprojection_error(nodes[i][1], b[:, :2], b[:, 2], Ps[cam_of[j]], 0.1, np.nan)
            else:
                D[i, j] = np.nan
    return D, dim_groups


# ----------------------------------------------------------------------------
# TR-1 / TR-2  DLT triangulation (+ optional post-optimise)
# ----------------------------------------------------------------------------
def dlt_point(projs, pts):
    """triangulate_point_from_multiple_views_linear, mv_math_util.py:215-240."""
    projs = np.asarray(projs, np.float64)
    pts = np.asarray(pts, np.float64)
    A = np.empty((2 * len(projs), 4))
    A[0::2] = pts[:, 0:1] * projs[:, 2, :] - projs[:, 0, :]
    A[1::2] = pts[:, 1:2] * projs[:, 2, :] - projs[:, 1, :]
    vh = np.linalg.svd(A, full_matrices=False)[2]
    return vh[3, :3] / vh[3, 3]


def triangulate_groups(projs, groups, min_score, post_optimize=False, n_max_iter=2):
    """triangulate_point_groups_from_multiple_views_linear, mv_math_util.py:152-212.
    projs (V,3,4); groups V x (J,3) -> (J,4) = x,y,z,mean score."""
    projs = np.asarray(projs, np.float64)
    groups = [np.asarray(g, np.float64) for g in groups]
    J = len(groups[0])
    out = np.empty((J, 4))
    for j in range(J):
        use = [v for v, g in enumerate(groups) if g[j, 2] >= min_score]
        if len(use) < 2:
            use = list(range(len(groups)))
        pts = np.array([groups[v][j] for v in use])
        out[j, :3] = dlt_point(projs[use], pts[:, :2])
        out[j, 3] = np.mean(pts[:, 2])
    if post_optimize:
        def _res(x):
            X = x.reshape((-1, 3))
            homo = np.concatenate([X, np.ones((X.shape[0], 1))], axis=-1).T
            d = []
            for v in range(len(projs)):
                h = projs[v] @ homo
                uv = (h[:2] / (h[2] + 1e-6)).T
                d.append(np.linalg.norm(uv - groups[v][:, :2], axis=-1) * groups[v][:, -1])
            return np.array(d).flatten()

        try:
            r = least_squares(_res, out[:, :3].ravel().copy(), max_nfev=n_max_iter)
            out[:, :3] = r.x.reshape(-1, 3)
        except Exception as e:  # mv_math_util.py:205-210 swallows + prints
            print(e)
    return out


# ----------------------------------------------------------------------------
# FK-1 / FK-2  quaternion Euler rotations and forward kinematics
# ----------------------------------------------------------------------------
def _quat_axis(angle, k):
    """Quaternions.from_angle_axis (Quaternions.py:442-447) for unit axis e_k,
    including its 1/(1+1e-10) normalisation."""
    q = np.zeros(angle.shape + (4,))
    q[..., 0] = np.cos(angle / 2.0)
    q[..., 1 + k] = (1.0 / (1.0 + 1e-10)) * np.sin(angle / 2.0)
    return q


def _quat_mul(q, r):
    """Quaternions.__mul__ (Quaternions.py:76-115): Hamilton product q (x) r."""
    q0, q1, q2, q3 = (q[..., k] for k in range(4))
    r0, r1, r2, r3 = (r[..., k] for k in range(4))
    return np.stack([r0 * q0 - r1 * q1 - r2 * q2 - r3 * q3,
                     r0 * q1 + r1 * q0 - r2 * q3 + r3 * q2,
                     r0 * q2 + r1 * q3 + r2 * q0 - r3 * q1,
                     r0 * q3 - r1 * q2 + r2 * q1 + r3 * q0], axis=-1)


def euler_to_rotmats(es):
    """from_euler(order='xyz', world=False).transforms(): R = Rx Ry Rz.
    Quaternions.py:449-462, :335-366."""
    es = np.asarray(es, np.float64)
    q = _quat_mul(_quat_axis(es[..., 0], 0), _quat_mul(_quat_axis(es[..., 1], 1), _quat_axis(es[..., 2], 2)))
    qw, qx, qy, qz = (q[..., k] for k in range(4))
    x2, y2, z2 = qx + qx, qy + qy, qz + qz
    xx, yy, wx = qx * x2, qy * y2, qw * x2
    xy, yz, wy = qx * y2, qy * z2, qw * y2
    xz, zz, wz = qx * z2, qz * z2, qw * z2
    m = np.empty(q.shape[:-1] + (3, 3))
    m[..., 0, 0] = 1.0 - (yy + zz)
    m[..., 0, 1] = xy - wz
    m[..., 0, 2] = xz + wy
    m[..., 1, 0] = xy + wz
    m[..., 1, 1] = 1.0 - (xx + zz)
    m[..., 1, 2] = yz - wx
    m[..., 2, 0] = xz - wy
    m[..., 2, 1] = yz + wx
    m[..., 2, 2] = 1.0 - (xx + yy)
    return m


def forward_kinematics(root, euler, blens_full_or_side, bone_dirs=None, side_map=SIDE_TO_FULL):
    """foward_kinematics, inverse_kinematics.py:176-199 -> (pos[18,3], G[18,4,4]).
    ``blens_full_or_side``: 11 side lengths (expanded through side_map,
    :115-117) or 18 full lengths when side_map is None (old pickle schema)."""
    if bone_dirs is None:
        bone_dirs = skeleton_constants()[0]
    bl = np.asarray(blens_full_or_side, np.float64)
    full = bl[side_map] if side_map is not None else bl
    offsets = bone_dirs * full[:, None]
    R = euler_to_rotmats(np.asarray(euler, np.float64).reshape(N_SKEL, 3))
    L = np.tile(np.eye(4), (N_SKEL, 1, 1))
    L[:, :3, :3] = R
    L[1:, :3, 3] = offsets[1:]
    if root is not None:
        L[0, :3, 3] = root
    G = L.copy()
    for j in range(1, N_SKEL):
        G[j] = G[SKEL_PARENTS[j]] @ L[j]
    pos = G[:, :, 3]
    return pos[:, :3] / pos[:, 3, None], G


# ----------------------------------------------------------------------------
# IK-1 .. IK-4
# ----------------------------------------------------------------------------
def add_mid_spine(kps17):
    """guess_mid_spine + _hack_add_midspine, inverse_kinematics.py:339-348,370-378:
    (17,3) -> (18,3)."""
    k = np.asarray(kps17, np.float64)
    mid_sh = 0.5 * (k[COCO_L_SHOULDER] + k[COCO_R_SHOULDER])
    mid_hip = 0.5 * (k[COCO_L_HIP] + k[COCO_R_HIP])
    sp = 0.5 * (mid_sh + mid_hip)
    sc = k[COCO_L_SHOULDER, -1] * k[COCO_R_SHOULDER, -1]
    sc *= k[COCO_L_HIP, -1] * k[COCO_R_HIP, -1]
    return np.concatenate([k, np.array([[sp[0], sp[1], sc]])], axis=0)


def ik_residual(root, euler, side_blens, obs, projs, bone_dirs=None):
    """IK-2 residual, inverse_kinematics.py:219-234 / :258-272.
    obs (V,16,3) already gathered through IK_OBS_IDX; projs (V,3,4)."""
    pos, _ = forward_kinematics(root, euler, side_blens, bone_dirs)
    X = pos[IK_SKEL_IDX]
    # per-view (3,4) @ (4,16) products in the reference's operand layout, so that
    # BLAS rounds identically (the truncated TRF solve amplifies 1-ulp changes)
    homo = np.concatenate([X, np.ones((len(X), 1), dtype=X.dtype)], axis=-1).T
    uv = []
    for v in range(len(projs)):
        h = projs[v] @ homo
        uv.append((h[:2] / (1e-5 + h[2])).T)
    uv = np.array(uv)
    return ((uv - obs[..., :2]) * obs[..., 2:3]).ravel()


def ik_stage1(obs, projs, root, euler, side_blens, max_nfev, bone_dirs=None):
    """solve_pose_reproj, inverse_kinematics.py:202-238 (x = root, euler)."""
    def fun(x):
        return ik_residual(x[:3], x[3:].reshape(-1, 3), side_blens, obs, projs, bone_dirs)
    r = least_squares(fun, np.concatenate([np.ravel(root), np.ravel(euler)]), max_nfev=max_nfev)
    return r.x[:3], r.x[3:].reshape(-1, 3), r


def ik_stage2(obs, projs, root, euler, side_blens, max_nfev, bone_dirs=None):
    """solve_pose_bone_lens_reproj, inverse_kinematics.py:241-277 (x = root, euler, side lengths)."""
    n3 = 3 + 3 * N_SKEL

    def fun(x):
        return ik_residual(x[:3], x[3:n3].reshape(-1, 3), x[n3:], obs, projs, bone_dirs)
    r = least_squares(fun, np.concatenate([np.ravel(root), np.ravel(euler), np.ravel(side_blens)]),
                      max_nfev=max_nfev)
    return r.x[:3], r.x[3:n3].reshape(-1, 3), r.x[n3:], r


def pose_solver_solve(cam_poses_2d, cam_projs, init=None, return_info=False):
    """PoseSolver(...).solve(), inverse_kinematics.py:351-433.
    cam_poses_2d: V x (17,3) COCO; cam_projs: V x (3,4); init: None (cold) or
    (root, euler[18,3], side_blens[11]).  -> ((root, euler, side_blens), joints[18,3])."""
    projs = np.asarray(cam_projs, np.float64)
    poses18 = [add_mid_spine(p) for p in cam_poses_2d]
    obs = np.array(poses18)[:, IK_OBS_IDX, :]
    bone_dirs, ref_side = skeleton_constants()
    if init is None:
        p3d = triangulate_groups(projs, poses18, 0.01, True)
        root = 0.5 * (p3d[COCO_L_HIP, :3] + p3d[COCO_R_HIP, :3])
        euler = np.zeros((N_SKEL, 3))
        blens = ref_side.copy()
        nfev = 50
    else:
        root, euler, blens = (np.asarray(a, np.float64) for a in init)
        nfev = 5
    r1, e1, res1 = ik_stage1(obs, projs, root, euler, blens, nfev, bone_dirs)
    r2, e2, b2, res2 = ik_stage2(obs, projs, r1, e1, blens, nfev, bone_dirs)
    joints, _ = forward_kinematics(r2, e2, b2, bone_dirs)
    if return_info:
        return (r2, e2, b2), joints, dict(init=(root, euler, blens), stage1=(r1, e1), res1=res1, res2=res2)
    return (r2, e2, b2), joints


def ik_residual_3d(root, euler, side_blens, target, bone_dirs=None):
    """_residual_step_joints_3d / _residual_root_angles_bone_lens of the 3-D-target variants
    (inverse_kinematics.py:295-299, :325-330): (joint - target[:, :3]) * target[:, 3:], target = obs_pose_3d[obs idx]."""
    joints, _ = forward_kinematics(root, euler, side_blens, bone_dirs)
    d = (joints[IK_SKEL_IDX, :] - target[:, :3]) * target[:, -1:]
    return d.flatten()


def ik_stage1_3d(obs_pose_3d, root, euler, side_blens, max_nfev, bone_dirs=None):
    """solve_pose, inverse_kinematics.py:280-307."""
    target = np.asarray(obs_pose_3d, np.float64)[IK_OBS_IDX, :]

    def fun(x):
        return ik_residual_3d(x[:3], x[3:].reshape(-1, 3), side_blens, target, bone_dirs)
    r = least_squares(fun, np.concatenate([np.ravel(root), np.ravel(euler)]), max_nfev=max_nfev)
    return r.x[:3], r.x[3:].reshape(-1, 3), r


def ik_stage2_3d(obs_pose_3d, root, euler, side_blens, max_nfev, bone_dirs=None):
    """solve_pose_bone_lens, inverse_kinematics.py:310-336."""
    target = np.asarray(obs_pose_3d, np.float64)[IK_OBS_IDX, :]
    n3 = 3 + 3 * N_SKEL

    def fun(x):
        return ik_residual_3d(x[:3], x[3:n3].reshape(-1, 3), x[n3:], target, bone_dirs)
    r = least_squares(fun, np.concatenate([np.ravel(root), np.ravel(euler), np.ravel(side_blens)]), max_nfev=max_nfev)
    return r.x[:3], r.x[3:n3].reshape(-1, 3), r.x[n3:], r


# ---- match_eig (mv_association.py:179-219): restated only to show why it cannot be pinned (tests/test_match_eig_cpu.py) ----
def _biparti(sim):
    """mv_association.py:179-184: 0/1 matrix of the maximum-weight bipartite assignment of a block."""
    from scipy.optimize import linear_sum_assignment
    rows, cols = linear_sum_assignment(sim, maximize=True)
    p = np.zeros_like(sim)
    p[rows, cols] = 1
    return p


def match_eig(s_mat, dim_group, return_eig=False):
    """mv_association.py:187-219.  The reference keeps the FIRST d columns of np.linalg.eig -- LAPACK geev's order, which is
    neither sorted nor invariant under a relabelling of the nodes -- so the result is a property of the LAPACK build, not of the
    algorithm.  Complex eigenpairs, where geev returns them, are kept as NumPy gives them."""
    s_mat = np.asarray(s_mat)
    dim_group = [int(v) for v in dim_group]
    n, d = len(dim_group) - 1, int(max(np.diff(dim_group)))
    z = np.zeros_like(s_mat)
    for i in range(n):
        for j in range(n):
            z[dim_group[i]:dim_group[i + 1], dim_group[j]:dim_group[j + 1]] = _biparti(
                s_mat[dim_group[i]:dim_group[i + 1], dim_group[j]:dim_group[j + 1]])
    lam, u = np.linalg.eig(z)
    lam_d, u_d = lam[:d], u[:, :d]
    with np.errstate(all="ignore"):
        u_d = u_d * np.sqrt(lam_d)
    z_out = np.zeros_like(s_mat)
    for i in range(n):
        for j in range(n):
            if i == j:
                continue
            with np.errstate(all="ignore"):
                zb = np.real(u_d[dim_group[i]:dim_group[i + 1]] @ u_d[dim_group[j]:dim_group[j + 1]].T)
            zb = np.nan_to_num(zb)
            zb[zb < 0] = 0
            z_out[dim_group[i]:dim_group[i + 1], dim_group[j]:dim_group[j + 1]] = _biparti(zb)
    mm = transform_closure(z_out)
    if return_eig:
        return mm, z_out, lam
    return mm, z_out
