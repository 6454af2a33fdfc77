"""Meetings of the paired stage-1 IK models by outcome (GPU; a library built with -DMVMC_PAIR_STATS, which reports them in the
fallback slot of ik_info):   MVMC_LIB_PATH=.../libmvmc_pstats.so python tools/pair_stats.py [frames]
Per solve: built both models / model made by the partner / built alone."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiview_motion_capture_amd import synth
from multiview_motion_capture_amd.pipeline import HotPath
from multiview_motion_capture_amd.tracker import run_chains_fused, check_chain_flags

F = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
data = synth.generate(F, 5, 4, 20260103, chain_len=16, frame_seed=20260103)
hp = HotPath(data["K"], data["Rt"])
kps = torch.from_numpy(data["kps25"]).cuda(); cnt = torch.from_numpy(data["counts"]).cuda()
out = run_chains_fused(hp, kps, cnt, 16, want_info=True)
torch.cuda.synchronize()
check_chain_flags(out)
info = out["ik_info"].cpu().numpy().reshape(-1, 8)
ok = np.isfinite(info[:, 7])
w = info[ok, 7].astype(np.int64)
b, f, a = w % 100, (w // 100) % 100, w // 10000
nf = info[ok, 1]
print("solves %d   meetings per solve: built both %.2f, made by the partner %.2f, alone %.2f" % (ok.sum(), b.mean(), f.mean(), a.mean()))
warm = nf <= 5
print("warm solves %d: both %.2f partner %.2f alone %.2f | cold %d: both %.2f partner %.2f alone %.2f" % (
    warm.sum(), b[warm].mean(), f[warm].mean(), a[warm].mean(), (~warm).sum(), b[~warm].mean(), f[~warm].mean(), a[~warm].mean()))
NP = out["ik_info"].shape[-2] if out["ik_info"].dim() >= 3 else 0
if NP:
    per = out["ik_info"].cpu().numpy().reshape(-1, NP, 8)
    for s in range(min(NP, 6)):
        x = per[:, s, 7]; m = np.isfinite(x)
        if m.sum() == 0: continue
        w = x[m].astype(np.int64)
        print("slot %d: solves %d  built both %.2f  by partner %.2f  alone %.2f" % (s, m.sum(), (w % 100).mean(), ((w // 100) % 100).mean(), (w // 10000).mean()))
