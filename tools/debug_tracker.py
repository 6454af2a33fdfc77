import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
from multiview_motion_capture_amd import device as dev
from multiview_motion_capture_amd.pipeline import HotPath
from multiview_motion_capture_amd.tracker import ChainTracker
si = np.load(os.path.join(ROOT, "tests/golden/shelf_inputs.npz")); g = np.load(os.path.join(ROOT, "tests/golden/shelf_tracker.npz"))
d = torch.device("cuda:0")
hp = HotPath(si["K"], si["Rt"], device=d)
kps17, cnt = dev.ingest(torch.from_numpy(si["kps25"]).to(d), torch.from_numpy(si["counts"].astype(np.int32)).to(d))
tr = ChainTracker(hp, 1, kps17.shape[2], t_max=8)
for fi in range(1, 301):
    out = tr.step(kps17[fi:fi+1].contiguous(), cnt[fi:fi+1].contiguous())
    meta = tr.meta[0, :int(tr.n_tracks[0])].cpu().numpy()
    exp = g["alive_after"][fi-1]; exp = exp[exp[:, 0] >= 0]
    ok = meta.shape == exp.shape and np.array_equal(meta, exp)
    if not ok or fi % 50 == 0:
        print(fi, "OK" if ok else "DIFF", "dev", meta.tolist(), "ref", exp.tolist(), "dead", int(tr.n_dead[0]), int(g["n_dead"][fi-1]),
              "status", out["status"][0].cpu().tolist()[:5], "n_new", int(out["n_new"][0]))
    if not ok and fi > 125: break
