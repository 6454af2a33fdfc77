"""Diagnostic (-DMVMC_ALS_PROFILE build): cycles per ALS iteration by phase."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiview_motion_capture_amd import synth, device as dev
from multiview_motion_capture_amd.pipeline import HotPath
data = synth.generate(256, 5, 4, 20260103)
hp = HotPath(data["K"], data["Rt"])
kps17, cnt = dev.ingest(torch.from_numpy(data["kps25"]).cuda(), torch.from_numpy(data["counts"]).cuda())
_, S = dev.affinity(kps17, cnt, hp.F, want_D=False)
Wp = torch.zeros((256, 28, 28), dtype=torch.float64, device="cuda"); Wp[:, :20, :20] = S.double()
gc = torch.cat([torch.zeros((256, 1), dtype=torch.int32, device="cuda"), cnt], 1).contiguous()
for name, W, g in (("f32 n=20 (NMAX 24)", S, cnt), ("f64 n=20 in ld 28 (NMAX 32)", Wp, gc)):
    res = dev.als_associate(W, g, g_max=4)
    torch.cuda.synchronize()
    lab = res["labels"].cpu().numpy(); it = res["iters"].cpu().numpy()
    ph = lab[:, -7:].astype(float)
    print(name, "iters mean", it.mean(), "cycles/iter by phase [X1, G+H, chain1, apply1+B, G2+H2, chain2, apply2+X+res]:", ph.mean(0).round(0), "total", ph.sum(1).mean().round(0))
