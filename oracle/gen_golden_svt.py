"""Golden vectors of match_svt (mv_association.py:321-411) by RUNNING THE REFERENCE (build container only; test infrastructure).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_svt.py

Inputs: the float32 affinity matrices S of the Shelf frames already recorded in tests/golden/shelf_spatial.npz (what
match_multiview_poses hands to the matcher, mv_association.py:414-436), the same matrices in float64, and two 64-node graphs (8 groups
of 8) of the C8 P8 size.  Outputs: X_bin, match_mat and the number of SVD calls (= iterations run).  Only data is written."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    import torch
    m = ref_shim.load_modules()
    g = np.load(os.path.join(OUT, "shelf_spatial.npz"))
    frames = sorted({int(k[1:].split("_")[0]) for k in g.files})
    cases = []
    for fi in frames:
        S, dim = g[f"f{fi}_S"], g[f"f{fi}_dim"]
        cases.append((f"shelf{fi}_f32", S.astype(np.float32), dim))
        cases.append((f"shelf{fi}_f64", S.astype(np.float64), dim))
    rng = np.random.default_rng(20260105)
    for k in range(2):   # block-structured 64-node graphs: 8 people seen in 8 views, noisy affinities
        n_g, p = 8, 8
        ident = np.concatenate([rng.permutation(p) for _ in range(n_g)])
        same = ident[:, None] == ident[None, :]
        S = np.where(same, rng.uniform(0.6, 1.0, (64, 64)), rng.uniform(0.0, 0.45, (64, 64)))
        S = 0.5 * (S + S.T)
        cases.append((f"c8p8_{k}_f64", S, np.arange(0, 65, 8)))
        cases.append((f"c8p8_{k}_f32", S.astype(np.float32), np.arange(0, 65, 8)))
    out = {}
    calls = [0]
    orig = torch.svd

    def counting_svd(*a, **k):
        calls[0] += 1
        return orig(*a, **k)
    torch.svd = counting_svd
    for name, S, dim in cases:
        calls[0] = 0
        mm, xb = m.assoc.match_svt(S.copy(), [int(v) for v in dim])
        out[f"{name}_S"], out[f"{name}_dim"] = S, np.asarray(dim, dtype=np.int64)
        out[f"{name}_x_bin"] = np.asarray(xb.numpy() if hasattr(xb, "numpy") else xb).astype(bool)
        out[f"{name}_match_mat"] = np.asarray(mm.numpy() if hasattr(mm, "numpy") else mm).astype(bool)
        out[f"{name}_svd_calls"] = np.int64(calls[0])
        print(name, S.dtype, S.shape, "svd calls", calls[0], "pairs", int(out[f"{name}_x_bin"].sum()))
    torch.svd = orig
    np.savez_compressed(os.path.join(OUT, "svt_cases.npz"), **out)
    print("wrote", os.path.join(OUT, "svt_cases.npz"), len(cases), "cases")


if __name__ == "__main__":
    main()
