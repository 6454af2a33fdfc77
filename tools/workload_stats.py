"""ALS iteration and IK evaluation statistics of the benchmark workload (config 4) from one run of the chain kernel."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiview_motion_capture_amd import synth
from multiview_motion_capture_amd.pipeline import HotPath
from multiview_motion_capture_amd.tracker import run_chains_fused
L, F = 16, 10000
data = synth.generate(F, 5, 4, 20260103, chain_len=L)
hp = HotPath(data["K"], data["Rt"])
out = run_chains_fused(hp, torch.from_numpy(data["kps25"]).cuda(), torch.from_numpy(data["counts"]).cuda(), L, want_info=True)
torch.cuda.synchronize()
it = out["als_iters"].cpu().numpy()          # (B, L)
for name, x in (("chain heads (match_spatial, n = 20, f32)", it[:, 0]), ("later frames (match_spatial_time, n = 24, f64)", it[:, 1:].ravel())):
    q = np.percentile(x, [5, 25, 50, 75, 95, 99])
    print("ALS iterations, %s: mean %.1f, percentiles 5/25/50/75/95/99 = %s, max %d" % (name, x.mean(), q.round(0).astype(int).tolist(), x.max()))
    h, e = np.histogram(x, bins=[0, 50, 100, 150, 200, 300, 500, 1001])
    print("   histogram", dict(zip(["<50", "50-99", "100-149", "150-199", "200-299", "300-499", ">=500"], (h / h.sum()).round(3).tolist())))
inf = out["ik_info"].cpu().numpy()           # (B, L, NP, 8): nfev1, cost1, .., nfev2 at [4], njev at [6], fallbacks at [7]
for name, sl in (("cold (head)", inf[:, 0]), ("warm", inf[:, 1:])):
    v = sl.reshape(-1, 8)
    v = v[~np.isnan(v[:, 1])]
    nf = v[:, 1] + v[:, 4]
    print("IK %s: %d solves, evaluations per solve mean %.2f (min %d, max %d), models per solve mean %.2f, fallback models per solve %.4f" % (
        name, len(v), nf.mean(), nf.min(), nf.max(), v[:, 6].mean(), v[:, 7].mean()))
