"""A/B of the launch-per-stage chain path (run_chains) and the persistent chain kernel (run_chains_fused)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiview_motion_capture_amd import synth
from multiview_motion_capture_amd.pipeline import HotPath
from multiview_motion_capture_amd.tracker import run_chains, run_chains_fused
L = 16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
data = synth.generate(B * L, 5, 4, 20260103, chain_len=L)
hp = HotPath(data["K"], data["Rt"])
kps = torch.from_numpy(data["kps25"]).cuda(); cnt = torch.from_numpy(data["counts"]).cuda()
res = {}
for name, fn in (("stages", run_chains), ("fused", run_chains_fused)):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = fn(hp, kps, cnt, L, want_info=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(name, "%.2f ms" % (dt * 1e3))
    res[name] = {k: v.cpu().numpy() for k, v in out.items() if isinstance(v, torch.Tensor)}
a, b = res["stages"], res["fused"]
for k in ("n_tracks", "meta", "n_dead"):
    print(k, "equal", np.array_equal(a[k], b[k]))
for k in ("params", "joints"):
    m = np.isfinite(a[k]) & np.isfinite(b[k])
    print(k, "bit-identical", np.array_equal(a[k], b[k]), "max abs diff %.3e" % np.abs(a[k] - b[k])[m].max())
ia, ib = a["ik_info"].reshape(-1, 8), b["ik_info"].reshape(-1, 8)
print("info NaN pattern equal", np.array_equal(np.isnan(ia), np.isnan(ib)), "info equal", np.array_equal(ia[~np.isnan(ia)], ib[~np.isnan(ib)]))
pc = b["phase_cycles"]
print("fused kernel, cycles per chain by phase (mean / max over chains), in M cycles:")
for k, name in enumerate(("graph", "ALS", "assign", "IK", "commit", "outputs", "total")):
    print("  %-8s %8.3f %8.3f" % (name, pc[:, k].mean() / 1e6, pc[:, k].max() / 1e6))
it = b["als_iters"]
print("ALS iterations per chain: mean %.0f max %.0f (sum over 16 frames); per frame mean %.1f max %d" % (it.sum(1).mean(), it.sum(1).max(), it.mean(), it.max()))
