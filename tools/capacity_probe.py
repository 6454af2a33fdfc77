"""Which capacity the occluded / false-detection workload exceeds (bit 0 of the capacity word): neither k_max nor v_max <= 8 removes it --
the closure of a frame with false detections can put more than eight poses into one cluster, and the IK holds eight views per solve.
  python tools/capacity_probe.py   (GPU)"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from multiview_motion_capture_amd import synth
from multiview_motion_capture_amd.pipeline import HotPath
from multiview_motion_capture_amd.tracker import run_chains_fused
d = torch.device("cuda", 0)
data = synth.generate(4096, 5, 4, 3, chain_len=16, occlusion=0.02, spurious=0.05)
hp = HotPath(data["K"], data["Rt"], device=d)
kps = torch.from_numpy(data["kps25"]).to(d); cnt = torch.from_numpy(data["counts"]).to(d)
for k_max, v_max in ((None, None), (8, None), (None, 8), (8, 8), (12, 8)):
    out = run_chains_fused(hp, kps, cnt, 16, k_max=k_max, v_max=v_max, want_info=True)
    torch.cuda.synchronize()
    print("k_max", k_max, "v_max", v_max, "flags", out["flags"][-4:].cpu().tolist(), "n_new max", int(out["n_tracks"].max()))
