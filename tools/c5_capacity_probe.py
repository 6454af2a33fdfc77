import sys, numpy as np, torch
sys.path.insert(0, ".")
from multiview_motion_capture_amd import synth
from multiview_motion_capture_amd.pipeline import HotPath
from multiview_motion_capture_amd.tracker import run_chains_fused, run_chains
F, C, P, L = 4096, 8, 8, 16
data = synth.generate(F, C, P, 20260105, chain_len=L)
d = torch.device("cuda:0")
hp = HotPath(data["K"], data["Rt"], device=d)
kps, cnt = torch.from_numpy(data["kps25"]).to(d), torch.from_numpy(data["counts"]).to(d)
for kmax, vmax in ((10, 8), (16, 8), (24, 8)):
    b = run_chains(hp, kps, cnt, L, k_max=kmax, v_max=vmax)
    torch.cuda.synchronize()
    ov = b["overflow"].cpu().numpy()
    n = b["n_tracks"].cpu().numpy()
    print("k_max", kmax, "v_max", vmax, "chains flagged", int((ov != 0).sum()), "of", len(ov), "words", np.unique(ov), "frames with 8 tracks %.4f" % (n == P).mean())
a = run_chains_fused(hp, kps, cnt, L, k_max=16)
torch.cuda.synchronize()
print("fused k_max 16 flags", a["flags"][-4:].cpu().tolist())
