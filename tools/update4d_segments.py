"""Where a call of MvTracker.update_4d spends its wall time on the Shelf sequence (GPU box): filling the inputs, issuing the launch,
waiting for the device (read_back), turning the tables into tracklets.   python tools/update4d_segments.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multiview_motion_capture_amd.common as common
import multiview_motion_capture_amd.inverse_kinematics as ik
import multiview_motion_capture_amd.motion_capture as mc
import multiview_motion_capture_amd.pose_def as pd
from multiview_motion_capture_amd import tracker as trk

with np.load(os.path.join(ROOT, "tests", "golden", "shelf_inputs.npz")) as z:
    si = {k: z[k] for k in ("K", "Rt", "kps25", "counts")}
calibs = [common.Calib.from_k_rt(si["K"][c], si["Rt"][c], (1032, 776)) for c in range(5)]
frames_all = []
for fi in range(1, 301):
    fr = []
    for c in range(5):
        poses = {}
        for p in range(int(si["counts"][fi, c])):
            coco = pd.conversion_openpose_25_to_coco(si["kps25"][fi, c, p])
            poses[p] = pd.Pose(pd.KpsFormat.COCO, coco[:, :2], coco[:, 2:], None)
        fr.append(mc.filter_bad_pose(common.FrameData(fi, poses, calibs[c], c + 1), 0.01, 4, 5))
    frames_all.append(fr)
skel = ik.load_skeleton()
acc = {"step_fused": 0.0, "read_back": 0.0, "upload": 0.0, "snapshot": 0.0}
for name, key in (("step_fused", "step_fused"), ("read_back", "read_back"), ("upload_inputs", "upload"), ("snapshot", "snapshot")):
    orig = getattr(trk.ChainTracker, name)
    def wrap(self, *a, _o=orig, _k=key, **kw):
        t0 = time.perf_counter()
        try:
            return _o(self, *a, **kw)
        finally:
            acc[_k] += time.perf_counter() - t0
    setattr(trk.ChainTracker, name, wrap)
for rep in range(3):
    for k in acc: acc[k] = 0.0
    tracker = mc.MvTracker(skel)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k, fr in enumerate(frames_all):
        tracker.update_4d(k + 1, fr, None)
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    rest = tot - sum(acc.values())
    print("pass %d: %.1f frames/s, %.3f ms per frame = upload %.3f + snapshot %.3f + step_fused (host side of the launch) %.3f + read_back (the wait "
          "for the device) %.3f + the rest of update_4d (inputs, tracklets) %.3f" % (rep, 300 / tot, tot / 300 * 1e3, acc["upload"] / 300 * 1e3,
          acc["snapshot"] / 300 * 1e3, acc["step_fused"] / 300 * 1e3, acc["read_back"] / 300 * 1e3, rest / 300 * 1e3))
