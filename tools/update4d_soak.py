"""Soak of the drop-in per-frame API (motion_capture.MvTracker.update_4d: filter_bad_pose -> the chain kernel on one frame -> the
Python tracklet objects, table widening / narrowing included) against the deterministic oracle tracker, on synthetic sequences
with occlusion and false detections (births, deaths, more live tracklets than the default table).  Test infrastructure; GPU box:
    python tools/update4d_soak.py > gpurun_out/update4d_soak.txt"""
import os
import sys
import multiprocessing
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# worker processes are SPAWNED, not forked: the parent has initialised the GPU, and a forked child would inherit its HIP state
SPAWN = multiprocessing.get_context("spawn")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle_soak import oracle_chain   # noqa: E402


def main():
    from multiview_motion_capture_amd import motion_capture as mc
    from multiview_motion_capture_amd import synth
    from multiview_motion_capture_amd.common import Calib
    n_frames = int(os.environ.get("SOAK_FRAMES", "48"))
    workloads = [(5, 4, 0.0, 0.0), (5, 4, 0.1, 0.5), (5, 6, 0.15, 0.6), (8, 8, 0.05, 0.3), (5, 8, 0.15, 0.9), (5, 7, 0.25, 0.8)]
    if os.environ.get("SOAK_WORKLOADS"):
        workloads = [workloads[int(i)] for i in os.environ["SOAK_WORKLOADS"].split()]
    seeds = [int(s) for s in os.environ.get("SOAK_SEEDS", "1 2 3").split()]
    with ProcessPoolExecutor(max_workers=int(os.environ.get("SOAK_WORKERS", "14")), mp_context=SPAWN) as pool:
        for C, P, occ, spur in workloads:
            datas = [synth.generate(n_frames, C, P, seed, chain_len=n_frames, occlusion=occ, spurious=spur) for seed in seeds]
            futs = [pool.submit(oracle_chain, (d["K"], d["Rt"], d["P"], d["kps25"].astype(np.float64), d["counts"])) for d in datas]
            frames = same = 0
            dd, first_bad, widest, errors = [], None, 0, []
            for seed, data, fut in zip(seeds, datas, futs):
                calibs = [Calib.from_k_rt(data["K"][c], data["Rt"][c], (1032, 776)) for c in range(C)]
                tracker = mc.MvTracker(p_max=P)
                rows, _, _ = fut.result()
                ok = True
                for tt, row in enumerate(rows):
                    if isinstance(row[0], str):
                        break
                    d_frames = [mc.filter_bad_pose(f, 0.01, 4, 5) for f in mc.frame_data_from_batch(tt, data["kps25"][tt].astype(np.float64), data["counts"][tt], calibs)]
                    try:
                        tracker.update_4d(tt, d_frames, None)
                    except Exception as exc:
                        errors.append((seed, tt, repr(exc)[:120]))
                        break
                    exp, jo, _ = row
                    got = [(t.track_id, t.state.value, t.hits, len(t)) for t in tracker.tracklets]
                    widest = max(widest, len(got))
                    frames += 1
                    if ok and got == [tuple(int(v) for v in r) for r in exp]:
                        same += 1
                        for s, t in enumerate(tracker.tracklets):
                            dj = float(np.abs(t.last_pose_3d.keypoints - jo[s]).max())
                            dd.append(dj)
                            if dj > 1e-6 and os.environ.get("SOAK_VERBOSE"):
                                print(f"    above 1e-6: seed {seed} frame {tt} slot {s} (id, state, hits, length) {exp[s].tolist()}: {dj:.1e} m", flush=True)
                    else:
                        if ok and first_bad is None:
                            first_bad = (seed, tt, got, exp.tolist())
                        ok = False
            dd = np.array(dd) if dd else np.array([np.nan])
            print(f"C{C} P{P} occlusion {occ} spurious {spur}: {len(seeds)} sequences of {n_frames} frames through update_4d: tables equal on {same} / "
                  f"{frames} frames (first difference: {first_bad}); most live tracklets {widest}; {len(dd)} tracklet-frames, joint difference "
                  f"median {np.nanmedian(dd):.1e} p99 {np.nanpercentile(dd, 99):.1e} max {np.nanmax(dd):.1e} m, above 1e-6: {int((dd > 1e-6).sum())}; "
                  f"exceptions: {errors}", flush=True)


if __name__ == "__main__":
    main()
