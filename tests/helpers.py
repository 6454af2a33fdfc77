"""Shared helpers of the parity tests (oracle side runs on the CPU, product side on the GPU)."""
import numpy as np

import oracle_np as o
from conftest import load_golden


def shelf_frames(frames):
    """(kps25 (F,C,P,25,3) f64, counts (F,C) i32, K, Rt, P) for the chosen Shelf frames."""
    g = load_golden("shelf_inputs.npz")
    return (g["kps25"][frames], g["counts"][frames].astype(np.int32), g["K"], g["Rt"], g["P"])


def oracle_ingest(kps25, counts):
    """Oracle IN-1/IN-2 on a padded batch -> (kps17 (F,C,P,17,3), counts (F,C))."""
    F, C, P = kps25.shape[:3]
    out = np.zeros((F, C, P, 17, 3))
    cnt = np.zeros((F, C), dtype=np.int32)
    for f in range(F):
        for c in range(C):
            k = 0
            for p in range(counts[f, c]):
                k17 = o.openpose25_to_coco17(kps25[f, c, p]) if kps25.shape[3] == 25 else kps25[f, c, p]
                if o.pose_is_good(k17):
                    out[f, c, k] = k17
                    k += 1
            cnt[f, c] = k
    return out, cnt


def frame_nodes(kps17_f, counts_f):
    """Compact node order of one frame: (points (n,17,2), scores (n,17), dim_group, pose index per node)."""
    C, P = kps17_f.shape[:2]
    pts, sc, dim, q = [], [], [0], []
    for c in range(C):
        for p in range(counts_f[c]):
            pts.append(kps17_f[c, p, :, :2])
            sc.append(kps17_f[c, p, :, 2])
            q.append(c * P + p)
        dim.append(dim[-1] + int(counts_f[c]))
    return np.array(pts).reshape(-1, 17, 2), np.array(sc).reshape(-1, 17), dim, q


def ulp_diff_f32(a, b):
    """|a-b| in units of float32 ulps of b (elementwise)."""
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    return np.abs(a.astype(np.float64) - b.astype(np.float64)) / np.spacing(np.abs(b).astype(np.float32)).astype(np.float64)


def postopt_step_noise_share(projs, poses18):
    """For triangulate(..., post_optimize=True) (mv_math_util.py:189-210; least_squares(max_nfev=2) = ONE trust-region trial step):
    the share of that step's squared length that lies along singular vectors of the Jacobian whose singular value is rounding noise
    (< 1e-10 s_max).  J is rank deficient (one residual per point and view for three unknowns per point), SciPy then normalises the
    step to |p| = Delta = |x0|, its Newton iteration drives alpha to ~ +-1e-20, and a LAPACK noise singular value of ~1e-16 then carries
    a coefficient s u^T f / (s^2 + alpha) ~ 1e6 x the genuine ones: the trial point is defined by rounding noise.  It is almost always
    rejected (cost goes up); where it is accepted, no other implementation can reproduce the reference's point."""
    import trf_np as t
    projs = np.asarray(projs, float)
    groups = [np.asarray(g, float) for g in poses18]

    def res(x):
        X = x.reshape((-1, 3))
        homo = np.concatenate([X, np.ones((X.shape[0], 1))], axis=-1).T
        d = []
        for v in range(len(projs)):
            h = projs[v] @ homo
            uv = (h[:2] / (h[2] + 1e-6)).T
            d.append(np.linalg.norm(uv - groups[v][:, :2], axis=-1) * groups[v][:, -1])
        return np.array(d).flatten()

    x0 = o.triangulate_groups(projs, groups, 0.01, False)[:, :3].ravel()
    f = res(x0)
    J = t.fd_jacobian(res, x0, f)
    U, s, Vt = np.linalg.svd(J, full_matrices=False)
    uf = U.T @ f
    _, alpha, _ = t.solve_tr_svd(J.shape[1], J.shape[0], uf, s, Vt.T, np.linalg.norm(x0), initial_alpha=0.0)
    with np.errstate(all="ignore"):
        c = np.where(s * uf != 0, s * uf / (s ** 2 + alpha), 0.0)
    noise = s < 1e-10 * s[0]
    return float((c[noise] ** 2).sum() / (c ** 2).sum())


def cpu_twin():
    """ctypes handle of the C++ CPU twin (oracle/cpu_twin/libmvmc_cpu.so, built by __graft_entry__.build() / make -C oracle)."""
    import ctypes
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "cpu_twin", "libmvmc_cpu.so")
    if not os.path.exists(path):
        raise RuntimeError("oracle/cpu_twin/libmvmc_cpu.so is not built: run make -C oracle")
    return ctypes.CDLL(path)


def twin_postopt_root(projs, poses18):
    """Root (midpoint of the post-optimised hips) from the HOST build of csrc/mvmc_trf_faithful.h -- the same source the device's
    TRF-faithful solver is compiled from."""
    import ctypes
    lib = cpu_twin()
    pose = np.ascontiguousarray(np.array(poses18, dtype=np.float64))
    Pm = np.ascontiguousarray(np.asarray(projs, dtype=np.float64))
    x = np.ascontiguousarray(o.triangulate_groups(Pm, list(pose), 0.01, False)[:, :3].ravel())
    out = np.zeros(4)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib.trf_check_postopt(p(pose), p(Pm), len(pose), 18, 2, p(x), p(out))
    X = x.reshape(18, 3)
    return 0.5 * (X[11] + X[12])


def als_oracle_job(args):
    """(S, dim) -> (X_bin, labels, iterations) of oracle_np.match_als: the unit of work of a process pool (the long ALS runs of config 5
    take the oracle ~40 ms each; tests/test_gpu_als_cap.py has 1,563 of them).  Workers are SPAWNED, import NumPy only and never touch the GPU."""
    S, dim = args
    mm, xb, it = o.match_als(S, dim, return_iters=True)
    return xb, o.cluster_labels(mm, len(S)), it

