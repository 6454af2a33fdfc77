"""ALS on the spatio-temporal graphs of the bench workload: iterations and cycles by phase.
Needs make -C multiview_motion_capture_amd/csrc prof-als; MVMC_LIB_PATH=multiview_motion_capture_amd/lib/libmvmc_prof.so.  The counters come back
through the label rows, so the tracker loses its tracklets on every other frame: read the frames that print phases."""
import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
from multiview_motion_capture_amd import synth, device as dev
from multiview_motion_capture_amd.pipeline import HotPath
from multiview_motion_capture_amd.tracker import ChainTracker
L, B = 16, 128
data = synth.generate(B * L, 5, 4, 20260103, chain_len=L)
hp = HotPath(data["K"], data["Rt"])
kps17, cnt = dev.ingest(torch.from_numpy(data["kps25"]).cuda(), torch.from_numpy(data["counts"]).cuda())
k4 = kps17.view(B, L, 5, 4, 17, 3); c4 = cnt.view(B, L, 5)
tr = ChainTracker(hp, B, 4)
its = []
for t in range(6):
    o = tr.step(k4[:, t].contiguous(), c4[:, t].contiguous(), want_debug=True)
    st = o["st"]
    it = st["iters"].cpu().numpy(); lab = st["labels"].cpu().numpy(); gc = o["group_counts"].cpu().numpy()
    if lab.shape[1] >= 28 and gc.sum(1).max() <= 24 and t > 0:   # als7: work / wait per phase, solver wave and a worker wave
        ph = np.nan_to_num(lab[it > 0][:, :28].astype(float).mean(0).round(0))
        names = ["X1|inv0-2", "rhsB|inv3-7", "applyB", "rhsA|form+inv", "applyA", "XZY|form"]
        print("frame", t, "iters mean %.1f" % it.mean(), " ".join(
            "%s: solver %d+%d worker %d+%d;" % (names[k], ph[2 * k], ph[2 * k + 1], ph[14 + 2 * k], ph[14 + 2 * k + 1]) for k in range(6)),
            "tail %d / %d; sum %d / %d" % (ph[12], ph[26], ph[:14].sum(), ph[14:].sum()))
        continue
    ph = lab[:, -7:].astype(float)
    print("frame", t, "n nodes", gc.sum(1).mean(), "gmax", gc.max(), "iters mean %.1f max %d" % (it.mean(), it.max()),
          "cycles/iter by phase:", ph[it > 0].mean(0).round(0), "sum", ph[it > 0].sum(1).mean().round(0))
