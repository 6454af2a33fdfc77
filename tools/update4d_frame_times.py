"""Per-frame wall time of MvTracker.update_4d over the 300 Shelf frames, split by what the frame contains (GPU box):
    python tools/update4d_frame_times.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from multiview_motion_capture_amd import motion_capture as mc
    from multiview_motion_capture_amd.common import Calib
    si = np.load(os.path.join(ROOT, "tests", "golden", "shelf_inputs.npz"))
    calibs = [Calib.from_k_rt(si["K"][c], si["Rt"][c], (1032, 776)) for c in range(5)]
    frames = [[mc.filter_bad_pose(f, 0.01, 4, 5) for f in mc.frame_data_from_batch(fi, si["kps25"][fi], si["counts"][fi], calibs)] for fi in range(1, 301)]
    best = None
    for rep in range(3):
        tracker = mc.MvTracker()
        ts, births = [], []
        for fi, fr in enumerate(frames):
            n0 = len(tracker.tracklets) + len(tracker.dead_tracklets)
            t0 = time.perf_counter()
            tracker.update_4d(fi + 1, fr, None)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            births.append(len(tracker.tracklets) + len(tracker.dead_tracklets) - n0)
        ts, births = np.array(ts) * 1e3, np.array(births)
        if best is None or ts.sum() < best[0].sum():
            best = (ts, births)
    ts, births = best
    b = births > 0
    print(f"300 Shelf frames through update_4d: {ts.sum():.0f} ms = {300e3 / ts.sum():.0f} frames/s; frames with a birth: {int(b.sum())} "
          f"({int(births.sum())} births), {ts[b].sum():.0f} ms ({100 * ts[b].sum() / ts.sum():.0f} % of the time; median {np.median(ts[b]):.2f} ms, max {ts[b].max():.2f} ms); "
          f"frames without: median {np.median(ts[~b]):.2f} ms, p90 {np.percentile(ts[~b], 90):.2f} ms")


if __name__ == "__main__":
    main()
