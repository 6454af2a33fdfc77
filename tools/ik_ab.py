"""A/B of the two IK kernels on the bench workload: wave-per-solve (+ flagged re-run) against workgroup-per-solve."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiview_motion_capture_amd import synth, _cabi  # noqa: E402
from multiview_motion_capture_amd.pipeline import HotPath  # noqa: E402
from multiview_motion_capture_amd.tracker import run_chains  # noqa: E402

L = 16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
data = synth.generate(B * L, 5, 4, 20260103, chain_len=L)
hp = HotPath(data["K"], data["Rt"])
kps = torch.from_numpy(data["kps25"]).cuda()
cnt = torch.from_numpy(data["counts"]).cuda()
lib = _cabi.load()
res = {}
for mode in (1, 0):
    lib.mvmc_debug_ik_mode(mode)
    for rep in range(2):
        ev = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = run_chains(hp, kps, cnt, L, want_info=True, events=ev)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    ik_ms = [a.elapsed_time(b) for a, b in ev]
    res[mode] = {k: v.cpu().numpy() for k, v in out.items()}
    print("mode", mode, "step %.2f ms; ik launches: cold %.2f, warm mean %.3f ms" % (dt * 1e3, ik_ms[0], np.mean(ik_ms[1:])))
a, b = res[1], res[0]
inf_a, inf_b = a["ik_info"].reshape(-1, 8), b["ik_info"].reshape(-1, 8)
ok = ~np.isnan(inf_a[:, 1])
print("solves", ok.sum(), "same NaN pattern", np.array_equal(np.isnan(inf_a), np.isnan(inf_b)))
print("n_tracks equal", np.array_equal(a["n_tracks"], b["n_tracks"]), "meta equal", np.array_equal(a["meta"], b["meta"]))
for name, col in (("nfev1", 1), ("status1", 2), ("nfev2", 4), ("status2", 5), ("njev", 6)):
    print(name, "equal fraction %.4f" % np.mean(inf_a[ok, col] == inf_b[ok, col]))
print("solves with an eigensolver fallback: workgroup kernel %.4f, wave kernel %.4f" % (np.mean(inf_a[ok, 7] > 0), np.mean(inf_b[ok, 7] > 0)))
fb = (inf_a[ok, 7] > 0) | (inf_b[ok, 7] > 0)
c_a, c_b = inf_a[ok, 3], inf_b[ok, 3]
rel = np.abs(c_a - c_b) / np.maximum(np.abs(c_a), 1e-300)
print("final cost rel diff: median %.2e p99 %.2e max %.2e" % (np.median(rel), np.percentile(rel, 99), rel.max()))
ja, jb = a["joints"], b["joints"]
m = np.isfinite(ja) & np.isfinite(jb)
dj = np.abs(ja - jb)[m]
print("joints abs diff: median %.2e p99 %.2e max %.2e" % (np.median(dj), np.percentile(dj, 99), dj.max()))

relf = rel[fb]
print("fallback solves only (%d): cost rel diff median %.2e p90 %.2e max %.2e" % (fb.sum(), np.median(relf), np.percentile(relf, 90), relf.max()))
