"""Seed soak of the chain kernel against the deterministic oracle tracker (test infrastructure; run on the GPU box):

    python tools/oracle_soak.py > gpurun_out/oracle_soak.txt

For several seeds, clean and with occlusion + false detections (births, deaths, ninth tracklets), every chain of a small batch goes through
mvmc_chain_run on the device and through tracker_np.OracleTracker driving trf_np.pose_solver_solve_clean on the host (a process per
chain).  Reported per workload: frames whose tracker table equals the oracle's, and the joint differences on those frames.  The tests hold
the same comparison on two fixed seeds (tests/test_gpu_synth_tracker.py); this is the wider net."""
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
L = 16


def oracle_chain(job):
    import oracle_np as o
    import tracker_np as tk
    import trf_np as t
    K, Rt, P, kps25, counts = job
    its = []
    orig = o.match_als

    def recording(W, dim, return_iters=False):      # the frame's ALS iteration count (tracker_np calls o.match_als once per frame)
        mm, xb, it = orig(W, dim, return_iters=True)
        its.append(it)
        return (mm, xb, it) if return_iters else (mm, xb)
    o.match_als = recording
    orc = tk.OracleTracker(K, Rt, P, solver=lambda poses, projs, init: t.pose_solver_solve_clean(poses, projs, init))
    rows = []
    for tt in range(kps25.shape[0]):
        views = []
        for c in range(kps25.shape[1]):
            poses = [o.openpose25_to_coco17(kps25[tt, c, p]) for p in range(int(counts[tt, c]))]
            views.append([q for q in poses if o.pose_is_good(q)])
        try:
            orc.update(tt, views)
        except Exception as exc:      # the reference raises on a frame without clusters (parse_match_result): the chain ends there
            rows.append(("raise", repr(exc)))
            break
        rows.append((np.array([[x.tid, x.state, x.hits, x.length] for x in orc.tracklets], dtype=np.int32).reshape(-1, 4),
                     np.array([x.joints for x in orc.tracklets]).reshape(-1, 18, 3), its[-1] if len(its) == tt + 1 else -1))
    o.match_als = orig
    return rows, orc.next_id, orc.n_dead


WORKLOADS = [(5, 4, 4, 0.0, 0.0), (5, 4, 4, 0.05, 0.2), (5, 4, 4, 0.15, 0.5), (8, 8, 1, 0.0, 0.0), (8, 8, 1, 0.05, 0.2)]


def main():
    workloads = WORKLOADS
    if os.environ.get("SOAK_WORKLOADS"):      # e.g. "3 4": indices into the list above
        workloads = [workloads[int(i)] for i in os.environ["SOAK_WORKLOADS"].split()]
    if os.environ.get("SOAK_CUSTOM"):         # e.g. "3,2,4,0.1,0.3,0.2,6.0;6,6,2,0,0": views, people, chains, occlusion, spurious[, drop, pixel sigma]
        workloads = [tuple(float(v) if "." in v else int(v) for v in w.split(",")) for w in os.environ["SOAK_CUSTOM"].split(";")]
    seeds = [int(s) for s in os.environ.get("SOAK_SEEDS", "1 2 3 4 5 6").split()]
    run(workloads, seeds, int(os.environ.get("SOAK_WORKERS", "14")))


def run(workloads, seeds, workers=14):
    """-> one dict per workload: frames, tables_equal, als_equal, als_capped, tracklet_frames, above_1e6 [(dj, hits)], worst"""
    import torch
    from multiview_motion_capture_amd import synth
    from multiview_motion_capture_amd.pipeline import HotPath
    from multiview_motion_capture_amd.tracker import repair_chains, run_chains_fused
    d = torch.device("cuda:0")
    results = []
    with ProcessPoolExecutor(max_workers=workers, mp_context=SPAWN) as pool:
        for wl in workloads:
            C, P, n_chains, occ, spur = wl[:5]
            extra = dict(zip(("drop", "pix_sigma"), wl[5:]))
            offenders = []
            frames = same = void = als_frames = als_same = als_cap = als_unexplained = repaired = 0
            dd, first_bad = [], None
            for seed in seeds:
                data = synth.generate(n_chains * L, C, P, seed, chain_len=L, occlusion=occ, spurious=spur, **extra)
                hp = HotPath(data["K"], data["Rt"], device=d)
                out = run_chains_fused(hp, torch.from_numpy(data["kps25"]).to(d), torch.from_numpy(data["counts"]).to(d), L, want_info=True)
                torch.cuda.synchronize()
                als_dev = out["als_iters"].cpu().numpy().reshape(-1)
                repaired_chains = set()
                if os.environ.get("SOAK_REPAIR") and int(out["void"].max()) != 0:
                    # the product path for chains beyond the layout's tables (a ninth tracklet, a graph of more than 32 nodes on the SMALL
                    # layout): tracker.repair_chains re-runs them stage by stage on the wide tier and splices their rows in
                    kps_t, cnt_t = torch.from_numpy(data["kps25"]).to(d), torch.from_numpy(data["counts"]).to(d)
                    repair_chains(hp, kps_t, cnt_t, out)
                    repaired_chains = set(out["repaired"].cpu().tolist())
                    repaired += len(repaired_chains)
                n_t, meta, joints = out["n_tracks"].cpu().numpy(), out["meta"].cpu().numpy(), out["joints"].cpu().numpy()
                vw = out["void"].cpu().numpy()
                k64 = data["kps25"].astype(np.float64)
                jobs = [(data["K"], data["Rt"], data["P"], k64[b * L:(b + 1) * L], data["counts"][b * L:(b + 1) * L]) for b in range(n_chains)]
                for b, (rows, next_id, n_dead) in enumerate(pool.map(oracle_chain, jobs)):
                    if vw[b]:
                        void += 1     # beyond the layout's tables: the repair tier's case, not compared here
                        continue
                    ok_chain = True
                    capped = False       # has this chain had a frame whose ALS ran into the cap (on either side)?
                    for tt, row in enumerate(rows):
                        if isinstance(row[0], str):
                            break
                        f = b * L + tt
                        frames += 1
                        exp, jo, it_o = row
                        if b in repaired_chains:
                            it_o = int(als_dev[f])       # (the repair tier does not report iteration counts: not compared)
                        als_frames += 1
                        als_same += int(it_o == als_dev[f])
                        als_cap += int(it_o >= 1000)
                        capped = capped or it_o >= 1000 or als_dev[f] >= 1000
                        als_unexplained += int(it_o != als_dev[f] and not capped)
                        if it_o != als_dev[f] and os.environ.get("SOAK_VERBOSE"):
                            print(f"    ALS count: seed {seed} chain {b} frame {tt}: oracle {it_o} device {als_dev[f]} (live tracklets {len(row[0])}, "
                                  f"chain has met the cap: {capped})", flush=True)
                        if ok_chain and n_t[f] == len(exp) and np.array_equal(meta[f, :len(exp)], exp):
                            same += 1
                            for s in range(len(exp)):
                                dj = float(np.abs(joints[f, s] - jo[s]).max())
                                dd.append(dj)
                                if dj > 1e-6:
                                    offenders.append((dj, int(exp[s][2])))
                                    tag = f"C{C}P{P}_occ{occ}_sp{spur}_seed{seed}_chain{b}"
                                    print(f"    above 1e-6: {tag} frame {tt} slot {s} (id, state, hits, length) {exp[s].tolist()}: {dj:.2e} m; the frame's ALS "
                                          f"iterations: oracle {it_o}, device {als_dev[f]}" + (" (AT THE CAP: the result of an unconverged iteration)" if it_o >= 1000 else ""),
                                          flush=True)
                                    if os.environ.get("SOAK_DUMP"):
                                        np.savez(os.path.join(os.environ["SOAK_DUMP"], tag + ".npz"), K=data["K"], Rt=data["Rt"], P=data["P"],
                                                 kps25=k64[b * L:(b + 1) * L], counts=data["counts"][b * L:(b + 1) * L],
                                                 dev_joints=joints[b * L:(b + 1) * L], dev_meta=meta[b * L:(b + 1) * L], dev_n=n_t[b * L:(b + 1) * L],
                                                 dev_params=out["out_params"].cpu().numpy()[b * L:(b + 1) * L] if "out_params" in out else out["params"].cpu().numpy()[b * L:(b + 1) * L])
                        else:
                            if ok_chain and first_bad is None:
                                first_bad = (seed, b, tt)
                            ok_chain = False      # (a chain's later frames follow from the first different table)
            dd = np.array(dd) if dd else np.array([np.nan])
            print(f"C{C} P{P} occlusion {occ} spurious {spur}{' ' + str(extra) if extra else ''}: {len(seeds)} seeds x {n_chains} chain(s) of {L}: tables equal on {same} / {frames} "
                  f"frames (first difference: {first_bad}; chains with a void word, not compared: {void}; chains that went through the repair tier: {repaired}); {len(dd)} tracklet-frames, joint "
                  f"difference median {np.nanmedian(dd):.1e} p90 {np.nanpercentile(dd, 90):.1e} p99 {np.nanpercentile(dd, 99):.1e} max {np.nanmax(dd):.1e} m; "
                  f"above 1e-6: {int((dd > 1e-6).sum())}; ALS iteration counts equal on {als_same} / {als_frames} frames ({als_cap} at the cap of 1000; different counts in a chain that has not met the cap: {als_unexplained})", flush=True)
            results.append(dict(workload=(C, P, n_chains, occ, spur), frames=frames, tables_equal=same, als_equal=als_same, als_capped=als_cap,
                                tracklet_frames=len(dd), above_1e6=offenders, worst=float(np.nanmax(dd))))
    return results


if __name__ == "__main__":
    main()
