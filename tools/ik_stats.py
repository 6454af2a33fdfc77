"""Evaluation / model / rejection statistics of the IK solves inside the chain kernel on the bench workload (GPU).

    python tools/ik_stats.py [frames] [views] [people]

Prints, per solve kind (cold / warm), the distribution of evaluations (nfev), models (njev) and rejected trials, the share of the
eigenbasis fallback, and the phase cycles of the kernel -- the numbers DESIGN.md section 6 quotes when it prices a design.
"""
import sys
import numpy as np
import torch

sys.path.insert(0, ".")
from multiview_motion_capture_amd import synth
from multiview_motion_capture_amd.pipeline import HotPath
from multiview_motion_capture_amd.tracker import run_chains_fused, check_chain_flags

F = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
C = int(sys.argv[2]) if len(sys.argv) > 2 else 5
P = int(sys.argv[3]) if len(sys.argv) > 3 else 4
L = 8
d = torch.device("cuda:0")
data = synth.generate(F, C, P, 20260103, chain_len=L, frame_seed=20260103)
hp = HotPath(data["K"], data["Rt"], device=d)
kps = torch.from_numpy(data["kps25"]).to(d)
counts = torch.from_numpy(data["counts"]).to(d)
out = run_chains_fused(hp, kps, counts, L, want_info=True)
torch.cuda.synchronize()
check_chain_flags(out)
info = out["ik_info"].cpu().numpy().reshape(-1, 8)
info = info[np.isfinite(info[:, 0])]
nfev = info[:, 1] + info[:, 4]
njev = info[:, 6]
cold = info[:, 1] > 6
for name, sel in (("cold", cold), ("warm", ~cold)):
    if not sel.any():
        continue
    nf, nj = nfev[sel], njev[sel]
    # per stage: trials = nfev - 1; a model per accepted trial except an accepted trial that exhausts the budget or stops
    rej = (nf - 2) - (nj - 2)
    print(f"{name}: {sel.sum()} solves | nfev mean {nf.mean():.2f} | njev mean {nj.mean():.2f} | trials - (models - 2) mean {rej.mean():.2f}"
          f" (upper bound of rejected trials; <= 2 of it are accepted last trials) | fallback solves {(info[sel, 7] > 0).mean():.3f}")
    print("   nfev hist", np.bincount(nf.astype(int))[:110].nonzero()[0][:20], "njev hist", np.bincount(nj.astype(int))[:40])
    print("   status s1", np.bincount(info[sel, 2].astype(int), minlength=5), "s2", np.bincount(info[sel, 5].astype(int), minlength=5))
ph = out["phase_cycles"].cpu().numpy()
print("phase cycles mean per chain (graph, als, assign, ik, commit, out, total, parts):", ph.mean(axis=0).round(0))
