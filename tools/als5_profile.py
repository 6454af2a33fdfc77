"""als5 (config 5, n = 72) cycles per iteration by phase.  Needs make -C multiview_motion_capture_amd/csrc prof-als and
MVMC_LIB_PATH=multiview_motion_capture_amd/lib/libmvmc_prof.so (the counters come back through the label rows)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
from multiview_motion_capture_amd import synth, device as dev
from multiview_motion_capture_amd.pipeline import HotPath
from multiview_motion_capture_amd.tracker import ChainTracker
L, B, C, P = 16, 64, 8, 8
data = synth.generate(B * L, C, P, 20260104, chain_len=L)
hp = HotPath(data["K"], data["Rt"])
kps17, cnt = dev.ingest(torch.from_numpy(data["kps25"]).cuda(), torch.from_numpy(data["counts"]).cuda())
k4 = kps17.view(B, L, C, P, 17, 3); c4 = cnt.view(B, L, C)
tr = ChainTracker(hp, B, P, t_max=8)
for t in range(3):
    o = tr.step(k4[:, t].contiguous(), c4[:, t].contiguous(), want_debug=True)
    for name in ("sp", "st"):
        it = o[name]["iters"].cpu().numpy(); lab = o[name]["labels"].cpu().numpy()
        if (it > 0).any():
            ph = lab[it > 0][:, :16].astype(float).mean(0).round(0)
            print("frame", t, name, "iters mean %.1f" % it[it > 0].mean(), "wave0:", ph[:8], "sum", ph[:8].sum(), "| wave3:", ph[8:], "sum", ph[8:].sum())
