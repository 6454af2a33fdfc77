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
