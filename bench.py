"""Benchmark of the hot path (BASELINE.json metric): frames/s for association + triangulation + IK
on synthetic (F frames x C views x P people x 25 joints) keypoints, one process per GPU.

    python bench.py [--gpus N --steps K --warmup W] [--frames F --views C --people P --workload full]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the whole hot path over this rank's F frames (inputs resident in HBM before
the timed region).  Frames shard across ranks as contiguous chain ranges (weak scaling: F per GPU is fixed) with no data-path
collective; each step ends with the multi-GPU tail of parallel.run_sharded on a communication stream: pack the live tracklets
(float32, ~0.5 KB per tracklet-frame) -> ONE all-gather (RCCL) -> stitch the identities across all chain boundaries on the device.
The tail runs at N = 1 too (the stitch over the shard's own chain boundaries), so the per-N values compare like with like.
Consecutive steps are independent batches and are issued on alternating HIP streams (--overlap N, default 3 for config 4: the next
launch's first frames fill the workgroup slots the slowest chains of the previous one leave idle; --overlap 1 = one step at a time).

`python bench.py --gpus N` started WITHOUT torch.distributed.run launches its N ranks itself, as fresh child processes, before this
process touches the GPU; under torch.distributed.run (RANK / WORLD_SIZE in the environment) it is one of the ranks.
Rank 0 prints ONE JSON line.
"""
import argparse
import contextlib
import json
import os
import sys
import time

# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and kernels of different streams that share a queue
# run in order.  This program keeps three (since round 6: six) steps in flight on streams of their own beside a communication stream, a check stream and
# the default stream: with four queues a step's pack kernel sat behind another step's chain kernel (1.55 ms per tail against 0.85 ms)
# and config 4 ran at 532 k frames/s instead of 548 k (same box, three runs each; 6, 8, 12, 16, 24 queues: the same 547 - 550 k).
# Read when the HIP runtime library is loaded, hence set before torch is imported; the environment overrides it.
# Only for the chain workloads (the default): the single-stream workloads keep the runtime's default (config 2 at BASELINE's literal
# 10 k frames, a 38 us launch, is slower with eight queues: 145 M against 266 M frames/s).
_wl = [a for i, a in enumerate(sys.argv) if a.startswith("--workload=") or (i and sys.argv[i - 1] == "--workload")]
if not _wl or _wl[-1].split("=")[-1] == "full":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_FRAME = lambda C, P, J=25: 12 * C * P * J + 4 * C * P + 16 * P * J + 488 * P
DEFAULT_NCCL_MAX_NCHANNELS = "4"   # see main(): RCCL beside the persistent chain kernel; chosen by the sweep in DESIGN.md section 7
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs x 16 fp64 lanes x 2 flop x 2.4 GHz (vector = fp64 MFMA rate on gfx950)
NOMINAL_CLOCK_HZ = 2.4e9
FP64_PEAK_TFLOPS = 78.6  # vector fp64


def cpu_baseline(data, n_frames, max_nfev):
    """Oracle (NumPy/SciPy restatement of the reference, 1 thread) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_np as o
    kps25, K, Rt, P = data["kps25"].astype(np.float64), data["K"], data["Rt"], data["P"]
    C, Pn = kps25.shape[1:3]
    t0 = time.perf_counter()
    Fm = o.pairwise_f_mats(K, Rt)
    n_solved = 0
    for f in range(n_frames):
        pts, poses, cam, dim = [], [], [], [0]
        for c in range(C):
            k = 0
            for p in range(Pn):
                k17 = o.openpose25_to_coco17(kps25[f, c, p])
                if o.pose_is_good(k17):
                    pts.append(k17[:, :2]); poses.append(k17); cam.append(c); k += 1
            dim.append(dim[-1] + k)
        D, S = o.geometry_affinity(np.array(pts), Fm, dim)
        mm, _ = o.match_als(S, dim)
        lab = o.cluster_labels(mm, len(pts))
        for k in range(lab.max() + 1):
            nodes = np.nonzero(lab == k)[0]
            if len(nodes) < 2:
                continue
            o.triangulate_groups(P[[cam[i] for i in nodes]], [poses[i] for i in nodes], 0.01, False)
            # PoseSolver.solve() cold start (DLT + 2 TRF stages)
            o.pose_solver_solve([poses[i] for i in nodes], [P[cam[i]] for i in nodes], None)
            n_solved += 1
    dt = time.perf_counter() - t0
    return dict(value=n_frames / dt, unit="frames/s", cores=1, kind="port",
                sample=f"{n_frames} frame(s) of the same synthetic workload ({n_solved} cold IK solves), "
                       f"oracle/oracle_np.py (NumPy + SciPy least_squares), {dt:.1f} s")


def _cpu_chain_worker(job):
    """One chain of L frames through the oracle tracker (runs in a worker process; NumPy/SciPy only)."""
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_np as o
    import tracker_np as tk
    kps25, K, Rt, P = job
    C, Pn = kps25.shape[1:3]
    tr = tk.OracleTracker(K, Rt, P)
    t0 = time.perf_counter()          # inside the worker, after the imports: interpreter / NumPy / SciPy start-up is not timed
    for f in range(kps25.shape[0]):
        views = []
        for c in range(C):
            poses = [o.openpose25_to_coco17(kps25[f, c, p]) for p in range(Pn)]
            views.append([p for p in poses if o.pose_is_good(p)])
        tr.update(f, views)
    n_cold = sum(1 for s in tr.solves if s[1])
    return n_cold, len(tr.solves) - n_cold, time.perf_counter() - t0


def cpu_baseline_chain(data, L, workers):
    """Oracle tracker (oracle/tracker_np.py: the reference's update_4d restated) over `workers` chains of L frames, one chain
    per worker process (the chains of the workload are independent, so this is how a CPU would run them in parallel)."""
    jobs = [(data["kps25"][w * L:(w + 1) * L].astype(np.float64), data["K"], data["Rt"], data["P"]) for w in range(workers)]
    t0 = time.perf_counter()
    if workers > 1:
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        env_old = os.environ.get("OMP_NUM_THREADS")
        os.environ["OMP_NUM_THREADS"] = "1"      # inherited by the spawned workers: one thread each
        try:
            with ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn")) as ex:
                res = list(ex.map(_cpu_chain_worker, jobs))
        finally:
            if env_old is None:
                os.environ.pop("OMP_NUM_THREADS", None)
            else:
                os.environ["OMP_NUM_THREADS"] = env_old
    else:
        res = [_cpu_chain_worker(jobs[0])]
    wall = time.perf_counter() - t0
    n_cold, n_warm = sum(r[0] for r in res), sum(r[1] for r in res)
    busy = max(r[2] for r in res)      # the workers run side by side: the job lasts as long as the slowest chain
    one = float(np.mean([r[2] for r in res]))
    return dict(value=workers * L / busy, unit="frames/s", cores=workers, kind="port",
                one_core_frames_per_s=L / one,
                sample=f"{workers} chain(s) of {L} frames of the same synthetic workload, one per worker process ({n_cold} cold + "
                       f"{n_warm} warm IK solves), oracle/tracker_np.py + oracle_np.py (NumPy + SciPy least_squares, one BLAS thread "
                       f"each); timed inside the workers after their imports: slowest chain {busy:.1f} s, mean {one:.1f} s "
                       f"({wall:.1f} s wall including worker start-up)")


def cpu_baseline_twin(data, L, n_frames_total):
    """The C++ CPU twin (oracle/cpu_twin: the reference's algorithm in C++17 -O3, SciPy's TRF restated literally, OpenMP over the
    independent chains) on a bounded sample of the same workload: all host cores, then one core."""
    import ctypes
    lib_path = os.path.join(ROOT, "oracle", "cpu_twin", "libmvmc_cpu.so")
    if not os.path.exists(lib_path):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(lib_path)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    C, Pn = data["kps25"].shape[1:3]
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    K, Rt = np.ascontiguousarray(data["K"]), np.ascontiguousarray(data["Rt"])

    def run(n_chains, threads):
        F = n_chains * L
        kps = np.ascontiguousarray(data["kps25"][:F])
        cnt = np.ascontiguousarray(data["counts"][:F].astype(np.int32))
        ns = np.zeros(F, dtype=np.int32)
        t0 = time.perf_counter()
        used = lib.mvmc_cpu_chain_run(ptr(K), ptr(Rt), ptr(kps), 0 if kps.dtype == np.float32 else 1, ptr(cnt), F, C, Pn, kps.shape[3], L, 8,
                                      50, 5, threads, None, None, None, None, None, ptr(ns))
        return F / (time.perf_counter() - t0), used, int(ns.sum()), time.perf_counter() - t0

    per_chain_s = 0.4 * (C * Pn / 20.0) ** 2             # rough single-core cost of a chain; keeps the sample near 10-20 s
    n_all = max(cores, min(n_frames_total // L, int(cores * 12.0 / per_chain_s)))
    n_one = max(1, min(n_frames_total // L, int(8.0 / per_chain_s)))
    v_all, used, solves, t_all = run(n_all, 0)
    v_one, _, _, t_one = run(n_one, 1)
    return dict(value=v_all, unit="frames/s", cores=used, kind="port", one_core_value=v_one,
                sample=f"C++ twin of the reference's algorithm (oracle/cpu_twin/mvmc_cpu.cpp: g++ -O3 -march=x86-64-v3 -fopenmp, SciPy's "
                       f"TRF with 2-point Jacobians and an SVD step restated literally): {n_all} chains of {L} frames of the same "
                       f"synthetic workload on {used} threads ({solves} IK solves, {t_all:.1f} s), then {n_one} chain(s) on one thread "
                       f"({t_one:.1f} s)")


def als_histogram(iters) -> dict:
    """the ALS / ADMM iteration histogram of one step (match_als, mv_association.py:275-309: data-dependent trip count,
    tol 1e-4, maxIter 1000) over the step's association graphs."""
    it = np.asarray(iters).reshape(-1)
    it = it[it > 0]
    if it.size == 0:
        return None
    edges = [1, 25, 50, 100, 150, 200, 300, 500, 1000]
    return {"graphs": int(it.size), "min": int(it.min()), "p50": float(np.percentile(it, 50)), "p90": float(np.percentile(it, 90)),
            "p99": float(np.percentile(it, 99)), "max": int(it.max()), "mean": float(it.mean()),
            "share_at_the_1000_cap": float((it >= 1000).mean()),
            "histogram": {f"[{lo}, {hi})": int(((it >= lo) & (it < hi)).sum()) for lo, hi in zip(edges, edges[1:])} | {"1000": int((it >= 1000).sum())}}


def kernel_sources_sha() -> str:
    """Hash of the HIP sources the library is built from (stamps profiles/pmc_traffic.json records; the same hash
    lib/BUILD_INFO.json carries and _cabi.load() checks)."""
    from multiview_motion_capture_amd import _buildinfo
    return _buildinfo.sources_sha()


def identities_carried(par, last, world):
    """What the stitch of one step carried: pairs matched across the chain boundaries INSIDE the shards (every rank's own, from the
    gathered messages' local sections), across the world - 1 boundaries BETWEEN shards, and the global identities that remain --
    with a continuous scene and everybody tracked: (chains - 1) x people pairs in all and `people` identities."""
    if last is None:
        return {}
    info = [int(v) for v in last["info"].cpu().tolist()]
    # word 1 of every message's `local` header (parallel.unpack_message's layout): only those eight words leave the device -- the
    # gathered messages are ~1 GB at eight ranks of config 5
    b_cap, row_cap = last["b_cap"], last["row_cap"]
    o_local = 8 + b_cap + b_cap * 2 * par.T_MSG * 56 + row_cap * 128
    within = int(last["messages"][:, o_local + 1].sum().item())
    return {"identities_carried_within_shard": within, "identities_carried_across_shards": info[3] - within,
            "global_identities": info[1], "shard_boundaries": world - 1}


def carries_against_ground_truth(out, data, L, Pn):
    """Rank 0's shard: every identity the stitch carried across one of the shard's chain boundaries, checked against the generator's
    ground truth -- the tracklet on either side of the boundary is the ground-truth person it lies on (mean joint distance), and a
    carry is right when both sides are the same person."""
    st = out.get("stitch")
    if st is None or "gid" not in st:
        return None
    gid = st["gid"].cpu().numpy()
    meta, n_t, jo = out["meta"].cpu().numpy(), out["n_tracks"].cpu().numpy(), out["joints"].cpu().numpy()
    gt = data["gt_joints"]
    F = meta.shape[0]
    B = F // L

    def people(f):   # {global identity: ground-truth person} of frame f's live tracklets
        b, n = f // L, int(n_t[f])
        if n == 0:
            return {}
        dm = np.linalg.norm(jo[f, :n, None] - gt[f][None], axis=-1).mean(axis=-1)      # (n, P)
        who = dm.argmin(axis=1)
        ok = dm.min(axis=1) < 0.25
        return {int(gid[b, int(meta[f, s, 0])]): int(who[s]) for s in range(n) if ok[s] and 0 <= meta[f, s, 0] < gid.shape[1]}

    carried = right = possible = 0
    for b in range(1, B):
        prev, nxt = people(b * L - 1), people(b * L)
        possible += len(set(prev.values()) & set(nxt.values()))
        for g, p in nxt.items():
            if g in prev:
                carried += 1
                right += int(prev[g] == p)
    return {"chain_boundaries_checked": B - 1, "people_on_both_sides": possible, "identities_carried": carried,
            "carried_to_the_right_person": right}


def _cabi_build_info():
    from multiview_motion_capture_amd import _cabi
    return _cabi.build_info()


def launch_ranks(n: int) -> int:
    """Start n ranks of this script (one per GPU, env as torch.distributed.run sets it) and wait; returns the exit code.
    Children are fresh interpreters (subprocess), so no process that has initialised the GPU is ever replaced or forked."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    import tempfile
    import time
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    # A rank that dies (no such device, a void step, an exception) must not leave the others waiting in a rendezvous or a collective
    # until their time-outs: the first non-zero exit ends the job -- the remaining ranks (these exact child processes) are terminated.
    codes = [None] * n
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            for r, p in enumerate(procs):
                if codes[r] is None:
                    p.terminate()
            for r, p in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[r] = p.wait()
            sys.stderr.write(f"bench.py: rank {bad[0]} exited with code {codes[bad[0]]}; the other ranks were stopped\n")
            break
        time.sleep(0.2)
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    return max(abs(c) for c in codes)


OTHER_CONFIGS = (
    ("config 1: the Shelf sequence (300 frames, 5 views) frame by frame through the drop-in MvTracker.update_4d",
     ["--workload", "shelf", "--steps", "2", "--warmup", "1"]),
    ("config 5: synthetic 25,008 frames/GPU, C=8, P=8 (the per-GPU shard of BASELINE's 200 k frames over 8 GPUs), full path",
     ["--views", "8", "--people", "8", "--frames", "25008", "--seed", "20260104", "--steps", "3", "--warmup", "1"]),
    ("config 3: synthetic 10 k frames, C=5, P=4: epipolar affinity + association + triangulation, every frame independent",
     ["--workload", "assoc_dlt", "--seed", "20260102", "--steps", "10", "--warmup", "2"]),
    ("config 2: synthetic 2 M frames, C=5, P=1: triangulation only (10 k generated frames tiled on the device)",
     ["--workload", "dlt", "--people", "1", "--frames", "2000000", "--tile-from", "10000", "--seed", "20260101", "--steps", "200",
      "--warmup", "20"]),      # (a launch lasts ~1 ms: 200 timed launches, so that the timed region is not the first 20 ms after an idle GPU)
    ("config 2 with float64 points out (the reference's output dtype, mv_math_util.py:152-187; 32 B per point instead of 16)",
     ["--workload", "dlt", "--people", "1", "--frames", "2000000", "--tile-from", "10000", "--seed", "20260101", "--steps", "200",
      "--warmup", "20", "--dlt-out", "f64"]),
)


def shelf_line(args, d):
    """BASELINE config 1: the Shelf sequence (tests/golden/shelf_inputs.npz: OpenPose keypoints of 5 views, frames 1 .. 300, the
    reference's own calibrations) through the mirrored driver loop -- parse -> filter_bad_pose -> MvTracker.update_4d, one frame after the
    other (motion_capture.py:1077-1111), the device doing association, triangulation and IK of each frame in one launch
    (mvmc_chain_run on a chain of one frame) and the host reading the frame's tracklets back, as the reference's API requires.  A step
    = the whole sequence with a fresh tracker.  The reference's NumPy path runs this at 1.7 frames/s."""
    import multiview_motion_capture_amd.common as common
    import multiview_motion_capture_amd.inverse_kinematics as ik
    import multiview_motion_capture_amd.motion_capture as mc
    import multiview_motion_capture_amd.pose_def as pd
    with np.load(os.path.join(ROOT, "tests", "golden", "shelf_inputs.npz")) as z:
        si = {k: z[k] for k in ("K", "Rt", "kps25", "counts")}        # (an NpzFile decompresses an array at every access)
    calibs = [common.Calib.from_k_rt(si["K"][c], si["Rt"][c], (1032, 776)) for c in range(5)]
    n_frames = min(300, si["kps25"].shape[0] - 1)
    frames_all = []
    for fi in range(1, n_frames + 1):          # the parsed, filtered FrameData of every frame: input preparation, not timed
        fr = []
        for c in range(5):
            poses = {}
            for p in range(int(si["counts"][fi, c])):
                coco = pd.conversion_openpose_25_to_coco(si["kps25"][fi, c, p])
                poses[p] = pd.Pose(pd.KpsFormat.COCO, coco[:, :2], coco[:, 2:], None)
            fr.append(mc.filter_bad_pose(common.FrameData(fi, poses, calibs[c], c + 1), 0.01, 4, 5))
        frames_all.append(fr)
    skel = ik.load_skeleton()

    def one_pass():
        tracker = mc.MvTracker(skel)
        for k, fr in enumerate(frames_all):
            tracker.update_4d(k + 1, fr, None)
        return tracker

    steps, warm = min(args.steps, 5), min(args.warmup, 1)
    for _ in range(warm):
        one_pass()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr = one_pass()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bpf = 12 * 5 * 8 * 25 + 4 * 5 * 8 + 16 * 4 * 25 + 488 * 4
    return {"metric": "frames/s (assoc+triangulate+IK), Shelf sequence frame by frame through MvTracker.update_4d",
            "value": n_frames * steps / dt, "unit": "frames/s", "n_gpus": 1, "steps": steps, "warmup": warm,
            "ms_per_step": dt / steps * 1e3, "ms_per_frame": dt / steps / n_frames * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "Shelf keypoints (tests/golden/shelf_inputs.npz)",
            "config": {"workload": f"Shelf, 5 cams, frames 1..{n_frames}, one frame per call of the drop-in MvTracker.update_4d (config 1)",
                       "frames": n_frames, "views": 5, "tracklets_alive_at_the_end": len(tr.tracklets), "dead": len(tr.dead_tracklets)},
            "roofline": {"bound": "hbm", "kernel": "chain_kernel<false> (one frame per launch)", "achieved": bpf * n_frames * steps / dt / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bpf * n_frames * steps / dt / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "note": "latency of one frame's dependent chain (one association, <= 6 solves) + a host round trip per frame: "
                                 "this leg measures the drop-in API, not the kernels' throughput"}}


def other_config_lines():
    """Runs the other BASELINE configurations as child processes of the headline run and returns their lines, trimmed."""
    import subprocess
    lines = []
    for name, argv in OTHER_CONFIGS:
        cmd = [sys.executable, os.path.abspath(__file__)] + argv + ["--sustain", "0", "--host-io", "0", "--cpu-frames", "0", "--no-other-configs"]
        t0 = time.perf_counter()
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
            line = json.loads(p.stdout.strip().splitlines()[-1])
        except Exception as exc:      # never at the cost of the headline line
            lines.append({"config": name, "error": repr(exc)[:300]})
            continue
        keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "ms_per_frame", "dtype", "config", "roofline",
                "tracker_events_per_step", "stages_ms", "accuracy", "collective", "als_iterations")
        rec = {"name": name, "command": "python bench.py " + " ".join(argv), **{k: line[k] for k in keep if k in line},
               "wall_s": time.perf_counter() - t0}
        lines.append(rec)
    return lines


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=10000, help="frames per GPU")
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--people", type=int, default=4)
    ap.add_argument("--workload", default="full", choices=["full", "assoc_dlt", "dlt", "shelf"],
                    help="full = config 4 (the headline); assoc_dlt = config 3 (affinity + ALS + DLT, every frame independent); "
                         "dlt = config 2 (triangulation only: ingest + DLT of one cluster per person; use --people 1); "
                         "shelf = config 1 (the Shelf sequence of tests/golden -- 300 frames, 5 views -- frame by frame through the drop-in "
                         "MvTracker.update_4d: a step = the whole sequence, one frame after the other, the host in the loop)")
    ap.add_argument("--nfev-cold", type=int, default=50)
    ap.add_argument("--nfev-warm", type=int, default=5)
    ap.add_argument("--chain-len", type=int, default=16,
                    help="frames per temporal chain (cold start at the head, warm after); 1 = every frame cold")
    ap.add_argument("--path", default="fused", choices=["fused", "stages"],
                    help="chain protocol: 'fused' = one persistent workgroup per chain in ONE launch (mvmc_chain_run); 'stages' = "
                         "one launch per stage and time step (ChainTracker.step)")
    ap.add_argument("--parts", type=int, default=0,
                    help="fused path: workgroups per chain (consecutive frame ranges handed over through device flags); "
                         "0 = one per frame of the chain (default), 1 = one persistent workgroup per chain")
    ap.add_argument("--groups", type=int, default=1,
                    help="chain groups advanced on separate HIP streams (association of one group overlaps IK of another)")
    ap.add_argument("--overlap", type=int, default=None,
                    help="steps in flight: consecutive steps (independent batches) are issued on N alternating HIP streams, so the "
                         "tail of one launch (its slowest chains, a fifth of the workgroup slots idle) is filled by the head of the "
                         "next; 1 = strictly one step after the other")
    ap.add_argument("--cpu-frames", type=int, default=1, help="frames of the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-workers", type=int, default=0,
                    help="worker processes of the CPU baseline on the chain protocol (0 = min(16, host cores))")
    ap.add_argument("--seed", type=int, default=20260103)
    ap.add_argument("--occlusion", type=float, default=0.0,
                    help="probability that a view misses a person in a frame (ragged counts, deaths and re-births); 0 = BASELINE workload")
    ap.add_argument("--spurious", type=float, default=0.0, help="probability that a slot freed by --occlusion holds a false detection instead (conditional)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the one all-gather: nccl = RCCL over xGMI (one rank per GPU); gloo only to "
                         "rehearse several ranks on a box with one GPU (host-staged)")
    ap.add_argument("--walk", default="continuous", choices=["chains", "continuous"],
                    help="synthetic scene: 'continuous' (default since round 6) = ONE smooth walk of the people over the whole sequence "
                         "(synth.scene_walk, SURVEY 8d), the same seed on every rank, rank r owning the r-th contiguous range of it: the "
                         "stitch has identities to carry across chain and shard boundaries; 'chains' restarts the people at every chain "
                         "head and gives every rank a scene of its own (rounds 1 - 5: nothing to carry)")
    ap.add_argument("--frames-total", type=int, default=0,
                    help="STRONG scaling: this many frames in all, split into contiguous ranges over the --gpus ranks (BASELINE config 5: "
                         "--views 8 --people 8 --frames-total 200064); overrides --frames (frames per GPU, weak scaling: the default)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--force-collective", action="store_true",
                    help="--gpus 1 only: initialise the backend at world size 1 and issue the step's all-gather through it (no world == 1 "
                         "short-cut) -- RCCL's kernels beside the resident chain workgroups on a one-GPU box; the N > 1 call path")
    ap.add_argument("--hand-over", default="auto", choices=["auto", "ticket", "static", "queue"],
                    help="fused path: how a chain's workgroups follow one another: ticket = (part, chain) from a ticket drawn at start "
                         "(no assumption about dispatch order; the default at every N), static = the same mapping by block index "
                         "(relies on in-order dispatch), queue = ready queue (no assumption, ~3 %% slower); auto = ticket")
    ap.add_argument("--host-io", type=int, default=20,
                    help="after the timed region: this many further steps with the host buffers of the boundary inside the bracket (inputs "
                         "from pinned host memory, results back to it), reported as 'host_io' beside 'value'; 0 = skip")
    ap.add_argument("--sustain", type=int, default=300,
                    help="after the timed region: this many further consecutive steps of the same command (capped at ~15 s of work), "
                         "reported as 'sustained' beside 'value'; 0 = skip")
    ap.add_argument("--tile-from", type=int, default=0,
                    help="generate this many frames on the host and tile them on the device up to --frames (config 2 at 2 M frames: the "
                         "generator would take minutes; the kernel reads every frame from HBM either way); 0 = generate all frames")
    ap.add_argument("--layout", choices=("auto", "big"), default="auto",
                    help="fused path: 'big' = the chain kernel's 512-thread layout (80-node graphs, 16 tracklet slots) also for views x people "
                         "<= 40 -- for geometries whose crowded frames exceed the SMALL layout's 32-node graphs (5 x 6 with everybody in view), "
                         "which otherwise go through the repair tier chain by chain")
    ap.add_argument("--dlt-out", default="f32", choices=["f32", "f64"],
                    help="--workload dlt: dtype of the triangulated points the one-pass kernel stores (float32 = SURVEY 8(d)'s I/O, one "
                         "16-byte store per point; float64 = mvmc_ingest_dlt's output, 32 bytes per point)")
    ap.add_argument("--other-configs", dest="other_configs", action="store_true", default=None,
                    help="after the headline region also time BASELINE config 5 (C8 P8, 25,008 frames, 3 steps), config 3 (C5 P4, association "
                         "+ triangulation, 10 steps) and config 2 (C5 P1, triangulation only, 2 M frames, 20 launches) as child runs of this script and report them under "
                         "'other_configs' (default: on for the default headline command at N = 1)")
    ap.add_argument("--no-other-configs", dest="other_configs", action="store_false")
    args = ap.parse_args()
    args.scaling = "weak"
    if args.frames_total:
        if args.frames_total % (args.gpus * args.chain_len):
            raise SystemExit("--frames-total must be a multiple of --gpus x --chain-len (whole chains per rank)")
        args.frames, args.scaling = args.frames_total // args.gpus, "strong"
    if args.other_configs is None:
        args.other_configs = (args.workload == "full" and (args.frames, args.views, args.people) == (10000, 5, 4)
                              and args.occlusion == 0.0 and args.path == "fused")
    if args.overlap is None:
        # steps in flight fill the tail of the chain kernel's launches; the memory-bound triangulation-only pass (config 2) has no
        # such tail, and overlapped launches would only make each of them last longer.  SMALL layout (four workgroups per CU: 1,024
        # slots against 10,000 workgroups per launch), eight hardware queues (see the top of this file), round 6's kernel, same box,
        # three alternations: 3 in flight 559.9 - 561.0 k frames/s, 5: 560.3 - 560.9 k, **6: 565.8 - 566.5 k**, 7: 539.6 k, 9: 549 -
        # 557 k (profiles/r06_ik_experiments.txt; round 5 had 2 and 4 at 530 - 534 k against 547 - 550 k at 3).  Six since round 6.
        # Config 5 (one workgroup per CU) does not care (109.9 k / 109.7 k) and keeps two.
        args.overlap = 1 if args.workload == "dlt" else (6 if args.views * args.people <= 40 and args.workload == "full" else 2)

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks as fresh child processes.  Nothing in this process has touched the GPU
        # (no HIP call, no torch.cuda call), and it never does: it only waits and passes rank 0's line through.
        raise SystemExit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE = {world}: launch N ranks for --gpus N")
    import torch.distributed as dist
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    d = torch.device("cuda", local_rank)
    if args.backend == "nccl" and (world > 1 or args.force_collective):
        # RCCL's channels are workgroups that need a slot beside the resident chain workgroups (768-1024 on the chip, spinning on
        # hand-over words): cap them.  The messages are small (20-100 MB per rank and step, overlapped with the next step), so the
        # collective does not need RCCL's default channel count; the cap is part of the record (collective.env)
        os.environ.setdefault("NCCL_MAX_NCHANNELS", DEFAULT_NCCL_MAX_NCHANNELS)
    if world > 1 or args.force_collective:
        if args.force_collective:
            if world != 1:
                raise SystemExit("--force-collective is the world-size-1 rehearsal of the N > 1 call path: --gpus 1")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                import socket
                sk = socket.socket()
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
                sk.close()
            from multiview_motion_capture_amd import parallel as par0
            par0.FORCE_COLLECTIVE = True
        kw = {}
        if args.backend == "nccl":
            kw["device_id"] = d
            try:       # RCCL's own stream at high priority too (its kernels share the GPU with 768 resident chain workgroups)
                opts = dist.ProcessGroupNCCL.Options()
                opts.is_high_priority_stream = True
                kw["pg_options"] = opts
            except Exception:
                pass
        try:
            dist.init_process_group(args.backend, rank=rank, world_size=world, **kw)
        except TypeError:      # (a torch without pg_options / device_id: the plain form)
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    res = run_workload(args, rank, world, d)
    if args.other_configs:
        import gc
        gc.collect()
        torch.cuda.empty_cache()       # (the headline workload's tensors died with run_workload's frame)
        if world == 1:
            # BASELINE configs 1, 5, 3 and 2 in the same driver-timed record: child runs of this script (fresh processes: their own
            # inputs, their own timed regions with the same barrier / synchronize bracket)
            res["other_configs"] = other_config_lines()
        else:
            # N > 1: BASELINE config 5 -- the configuration BASELINE defines BY its scaling curve (200 k frames, C8 P8 over 8 GPUs =
            # 25,008 frames per GPU) -- in the same processes and the same process group, every rank its own shard, the same barrier +
            # synchronize bracket and max over ranks; the single-GPU configurations (1, 2, 3) are in the N = 1 line
            a5 = argparse.Namespace(**vars(args))
            a5.views, a5.people, a5.frames, a5.seed, a5.steps, a5.warmup, a5.sustain, a5.cpu_frames = 8, 8, 25008, 20260104, 3, 1, 0, 0
            a5.host_io = 0
            t5 = time.perf_counter()
            r5 = run_workload(a5, rank, world, d)
            if rank == 0:
                keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "per_rank_ms_per_step", "dtype", "config",
                        "roofline", "tracker_events_per_step", "stages_ms", "accuracy", "collective", "als_iterations")
                res["other_configs"] = [{"name": OTHER_CONFIGS[1][0] + f", on {world} GPUs (all ranks of this run, one all-gather per step)",
                                         "command": "in-process: --views 8 --people 8 --frames 25008 --seed 20260104 --steps 3 --warmup 1",
                                         **{k: r5[k] for k in keep if k in r5}, "wall_s": time.perf_counter() - t5}]
    if rank == 0:
        print(json.dumps(res))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def run_workload(args, rank, world, d):
    """One workload on this rank (inside an initialised process group when world > 1): inputs -> warm-up -> the timed region -> the
    record (rank 0; None elsewhere).  Called once for the headline and, at N > 1, once more for BASELINE config 5."""
    import torch.distributed as dist
    if args.workload == "shelf":
        if world != 1:
            raise SystemExit("--workload shelf is a single-process sequence (the reference's driver loop): --gpus 1")
        return shelf_line(args, d) if rank == 0 else None
    from multiview_motion_capture_amd import synth
    from multiview_motion_capture_amd.pipeline import HotPath
    from multiview_motion_capture_amd import parallel as par
    from multiview_motion_capture_amd.tracker import check_chain_flags, repair_chains, run_chains, run_chains_fused

    F, C, Pn = args.frames, args.views, args.people
    # same cameras on every rank (seed); the frames: see --walk
    L = args.chain_len
    if F % L:
        raise SystemExit("--frames must be a multiple of --chain-len")
    F_gen = F
    if args.tile_from:
        if args.workload != "dlt" or F % args.tile_from or args.tile_from % L:
            raise SystemExit("--tile-from: only with --workload dlt (independent frames), and it must divide --frames")
        F_gen = args.tile_from
    if args.walk == "continuous":
        # one scene (the same seed everywhere), this rank's contiguous range of it.  The scene is made of segments of at most 32,768
        # frames (synth.scene_walk: a segment is a function of seed, index and length), rank r owning the segments r k .. r k + k - 1:
        # the generator's working set stays bounded (a 200 k-frame C8 P8 range in one piece needs ~30 GB of host memory), and BASELINE
        # config 5's strong-scaling form (--frames-total 200064) is the SAME eight segments of 25,008 frames at N = 1, 2, 4 and 8.
        k_seg = -(-F_gen // 32768)
        while F_gen % (k_seg * L):
            k_seg += 1
        parts = [synth.generate(F_gen // k_seg, C, Pn, args.seed, shuffle=args.workload != "dlt", occlusion=args.occlusion,
                                spurious=args.spurious, walk="scene", segment=rank * k_seg + j) for j in range(k_seg)]
        data = dict(parts[0])
        if k_seg > 1:
            for key in ("kps25", "counts", "gt_joints", "gt_order"):
                data[key] = np.concatenate([p[key] for p in parts], axis=0)
        del parts
    else:
        data = synth.generate(F_gen, C, Pn, args.seed, chain_len=L, frame_seed=args.seed + 1000 * rank,
                              shuffle=args.workload != "dlt", occlusion=args.occlusion, spurious=args.spurious)
    hp = HotPath(data["K"], data["Rt"], device=d)
    kps = torch.from_numpy(data["kps25"]).to(d)
    counts = torch.from_numpy(data["counts"]).to(d)
    if F_gen != F:
        kps = kps.repeat(F // F_gen, 1, 1, 1, 1).contiguous()
        counts = counts.repeat(F // F_gen, 1).contiguous()
    with_ik = args.workload == "full"
    dlt_members = None
    if args.workload == "dlt":      # cluster (f, p) = person slot p of every view (the generator's unshuffled order)
        f_i = torch.arange(F, device=d, dtype=torch.int32)[:, None, None]
        p_i = torch.arange(Pn, device=d, dtype=torch.int32)[None, :, None]
        c_i = torch.arange(C, device=d, dtype=torch.int32)[None, None, :]
        dlt_members = ((f_i * C + c_i) * Pn + p_i).reshape(F * Pn, C).contiguous()

    ev = {k: [] for k in ("assoc", "tri", "ik", "total")}

    ik_events, als_events, hand_over_flags = [], [], []
    kern_events = []   # every launch of the chain kernel in this process: (timed?, start, end)

    def step(timed, kps=kps, counts=counts):
        if with_ik and L > 1:
            # temporal protocol: chains of L frames, MvTracker.update_4d semantics
            e = [torch.cuda.Event(enable_timing=True) for _ in range(2)] if timed else None
            if timed: e[0].record()
            if args.path == "fused":
                kev = []
                out = run_chains_fused(hp, kps, counts, L, nfev_cold=args.nfev_cold, nfev_warm=args.nfev_warm, want_info=timed,
                                       parts=args.parts or None, kernel_events=kev,
                                       hand_over=None if args.hand_over == "auto" else args.hand_over,
                                       force_big=args.layout == "big")
                kern_events.append((timed, kev[0][0], kev[0][1]))
            else:
                out = run_chains(hp, kps, counts, L, nfev_cold=args.nfev_cold, nfev_warm=args.nfev_warm,
                                 events=ik_events if timed else None, want_info=timed, n_groups=args.groups,
                                 als_events=als_events if timed else None)
            if timed:
                e[1].record()
                ev["total"].append((e[0], e[1]))
            info = out.pop("ik_info", None)
            phase = out.pop("phase_cycles", None)
            out["als_it"] = out.pop("als_iters", None)
            out["info"] = info
            out["phase"] = phase
            return out
        if args.workload == "dlt":
            # config 2: every person's views are one cluster (no association): ingest + DLT
            from multiview_motion_capture_amd import device as dev
            # one pass (mvmc_ingest_dlt): the 17-joint tensor stays in LDS; --path stages = the two kernels mvmc_ingest + mvmc_dlt
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if timed else None
            if timed: e[0].record()
            if args.path == "fused":
                if timed: e[1].record()
                # float32 keypoints in, float32 points out: SURVEY 8(d)'s I/O for this configuration (12 C P J + 16 P J bytes per
                # frame) -- the float64 arithmetic of the reference, each point rounded once at a 16-byte store; --dlt-out f64 = the
                # float64 output the parity tests compare bit for bit
                pts = dev.ingest_dlt(kps, counts, hp.P, dlt_members.view(F, Pn, C),
                                     out_dtype=torch.float32 if args.dlt_out == "f32" else torch.float64)
            else:
                k17, c17 = dev.ingest(kps, counts)
                if timed: e[1].record()
                pts = dev.dlt(k17, hp.P, dlt_members)
            if timed:
                e[2].record()
                ev["assoc"].append((e[0], e[1])); ev["tri"].append((e[1], e[2]))
            return dict(pts3d=pts)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if timed else None
        if timed: e[0].record()
        assoc = hp.associate(kps, counts)
        if timed: e[1].record()
        tri = hp.triangulate(assoc)
        if timed: e[2].record()
        out = dict(labels=assoc["labels"], pts3d=tri["pts3d"], als_it=assoc.get("iters"))
        if with_ik:
            out.update(hp.solve_cold(assoc, tri, args.nfev_cold))
        if timed: e[3].record()
        if timed:
            ev["assoc"].append((e[0], e[1])); ev["tri"].append((e[1], e[2])); ev["ik"].append((e[2], e[3]))
        return out

    torch.cuda.synchronize()   # inputs and calibration tables were made on the default stream; the steps run on side streams
    streams = [torch.cuda.Stream(device=d) for _ in range(args.overlap)] if args.overlap > 1 else None

    sharded = with_ik and L > 1       # the temporal protocol ends every step with pack -> all-gather -> stitch
    # (high priority: the tail's small kernels -- and at N > 1 the collective -- get the next free workgroup slot instead of queueing
    # behind the next step's chain workgroups; config 4 at N = 1: 491.5 k -> 494.3 k frames/s, config 5 unchanged)
    comm = torch.cuda.Stream(device=d, priority=int(os.environ.get("MVMC_COMM_PRIORITY", "-1"))) if sharded else None
    tail_events, stitched = [], []

    fused_chain = sharded and args.path == "fused"
    repaired_per_step = []     # chains per step that went through the repair tier (tracker.repair_chains)

    def stream_ctx(i):
        return torch.cuda.stream(streams[i % len(streams)]) if streams is not None else contextlib.nullcontext()

    # --host-io: the same steps with the boundary's host buffers inside the timed region -- the shard's keypoints and counts from pinned
    # host memory before a step, its tables (or points) back to pinned host memory after it, on the step's own stream, so that the
    # copies of one step overlap the kernels of the others.  Reported as "host_io", never as "value".
    io_in, io_out, io_bytes = None, {}, [0, 0]
    if args.host_io > 0 and F_gen == F:
        kps_h, counts_h = torch.from_numpy(data["kps25"]).pin_memory(), torch.from_numpy(data["counts"]).pin_memory()
        io_in = [(torch.empty_like(kps), torch.empty_like(counts)) for _ in range(max(1, args.overlap))]
        io_bytes[0] = kps_h.numel() * kps_h.element_size() + counts_h.numel() * counts_h.element_size()

    def issue(i, timed, io=False):
        """Launch the compute of step i (asynchronous); finish() completes the step."""
        with stream_ctx(i):
            if io:
                kb, cb = io_in[i % len(io_in)]      # (the slot's previous user ran on this same stream)
                kb.copy_(kps_h, non_blocking=True)
                cb.copy_(counts_h, non_blocking=True)
                return (i, timed, step(timed, kb, cb), True)
            return (i, timed, step(timed), False)

    def to_host(i, out):
        """the step's results to pinned host memory (its own stream, behind its kernels)"""
        n = 0
        for k in ("params", "joints", "meta", "n_tracks", "pts3d", "labels"):
            v = out.get(k)
            if isinstance(v, torch.Tensor):
                h = io_out.get((i % len(io_in), k))
                if h is None or h.shape != v.shape:
                    h = io_out[(i % len(io_in), k)] = torch.empty(v.shape, dtype=v.dtype, pin_memory=True)
                h.copy_(v, non_blocking=True)
                n += v.numel() * v.element_size()
        io_bytes[1] = n

    check_stream = torch.cuda.Stream(device=d) if sharded else None

    def finish(rec):
        """The end of a step's compute: pack (+ the shard's own stitch) -> all-gather -> stitch on the communication stream, behind an
        event -- no device word is read here.  The chain kernel's validity words (hand-over time-out, capacities) travel in the message,
        so every rank learns of a void step from the gathered messages; settle() looks at them a few steps later."""
        i, timed, out, io = rec
        with stream_ctx(i):
            if io:
                to_host(i, out)
            if not sharded:
                return out
            res = par.run_sharded(lambda: out, L, (F // L) * world, rank, world, rows_per_frame=min(Pn + 1, par.T_MSG), comm_stream=comm,
                                  timing=timed, t_msg=par.T_MSG)
            res["step"], res["timed"] = i, timed
            return res

    def settle(res):
        """A finished step's verdict, read once its tail is done (the host is `overlap` steps ahead by then: one small read on an
        idle stream).  Void on some rank -> every rank has seen it: the rank concerned runs its repair tier (tracker.repair_chains: the
        chains that outgrew the chain kernel's tables again through the per-stage entry points), then all ranks repeat the step's tail.
        A step that stays void ends the run."""
        if not sharded:
            return res
        out = res["local"]
        with torch.cuda.stream(check_stream):
            try:
                par.check_stitch_info(res)
                if fused_chain:
                    repaired_per_step.append(0)
            except RuntimeError as exc:
                if not (fused_chain and "void" in str(exc)):
                    raise SystemExit(f"bench.py: step {res['step']} is void: {exc}")
                try:
                    repaired_per_step.append(repair_chains(hp, kps, counts, out, nfev_cold=args.nfev_cold, nfev_warm=args.nfev_warm))
                    check_chain_flags(out)
                    timed = res["timed"]
                    res = par.run_sharded(lambda: out, L, (F // L) * world, rank, world, rows_per_frame=min(Pn + 1, par.T_MSG), comm_stream=comm,
                                          timing=timed, t_msg=par.T_MSG)
                    res["timed"] = timed
                    par.check_stitch_info(res)
                except (ValueError, RuntimeError) as exc2:
                    raise SystemExit(f"bench.py: step {res.get('step')} is void: {exc2}")
        out.pop("_keepalive", None)
        if res["timed"] and res.get("tail_events"):
            tail_events.append(res["tail_events"])
        stitched.append(res)
        if len(stitched) > 3:
            stitched.pop(0)
        out["stitch"] = res
        return out

    def run_steps(n, timed, io=False):
        """n steps, args.overlap of them in flight: step i + 1 is launched before step i's tail is queued, and a step is settled
        (its validity read) `overlap` steps after its tail was queued -- the only place the host waits for the device."""
        pending, tails, last = [], [], None
        for i in range(n):
            pending.append(issue(i, timed, io))
            if len(pending) >= max(1, args.overlap):
                tails.append(finish(pending.pop(0)))
            while len(tails) > max(1, args.overlap):
                last = settle(tails.pop(0))
        while pending:
            tails.append(finish(pending.pop(0)))
        while tails:
            last = settle(tails.pop(0))
        return last

    run_steps(args.warmup, False)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run_steps(args.steps, True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    per_rank_ms = [dt / args.steps * 1e3]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=d if args.backend == "nccl" else "cpu")
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        per_rank_ms = [float(x.item()) / args.steps * 1e3 for x in parts]
        dt = max(float(x.item()) for x in parts)
    for res in stitched:
        par.check_stitch_info(res)     # a message overflowed / too many identities in a chain: the step's stitch is void

    # for the record: one step on its own, nothing else in flight (untimed region)
    serial_ms = None
    if args.overlap > 1:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step(False)
        e1.record()
        torch.cuda.synchronize()
        serial_ms = e0.elapsed_time(e1)
    # the same command, sustained: further consecutive steps (same barrier + synchronize bracket, max over ranks)
    sustained = None
    if args.sustain > 0:
        n_sus = max(args.steps, min(args.sustain, int(15.0 / (dt / args.steps))))
        if world > 1:
            t = torch.tensor([n_sus], dtype=torch.int64, device=d if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            n_sus = int(t.item())
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(n_sus, False)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dts = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dts], dtype=torch.float64, device=d if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dts = float(t.item())
        sustained = {"value": F * world * n_sus / dts, "unit": "frames/s", "steps": n_sus, "ms_per_step": dts / n_sus * 1e3}
        for res in stitched:
            par.check_stitch_info(res)
    host_io = None
    if io_in is not None:
        n_io = args.host_io
        run_steps(min(3, n_io), False, io=True)      # the pinned result buffers are made here
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(n_io, False, io=True)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dti = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dti], dtype=torch.float64, device=d if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dti = float(t.item())
        host_io = {"value": F * world * n_io / dti, "unit": "frames/s", "steps": n_io, "ms_per_step": dti / n_io * 1e3,
                   "h2d_bytes_per_step": io_bytes[0], "d2h_bytes_per_step": io_bytes[1],
                   "note": "the same steps with every step's keypoints + counts copied from pinned host memory and its tables (or points) "
                           "copied back to pinned host memory inside the bracket, on the step's stream (PCIe-inclusive; not `value`)"}
        for res in stitched:
            par.check_stitch_info(res)
    # the one collective of a step, as this run saw it (rank 0's events; every rank issues the same call)
    collective = None
    if sharded:
        last = stitched[-1] if stitched else None
        gather_ms = [e[1].elapsed_time(e[2]) for e in tail_events] if tail_events else []
        collective = {"backend": (dist.get_backend() if dist.is_initialized() else "none (world size 1: the message is its own gather)"),
                      "world_size": dist.get_world_size() if dist.is_initialized() else 1,
                      "forced_at_world_size_1": bool(par.FORCE_COLLECTIVE),
                      "all_gathers_per_step": 1,
                      "messages_gathered": int(last["messages"].shape[0]) if last is not None else None,
                      "message_bytes": int(last["message"].numel() * last["message"].element_size()) if last is not None else None,
                      "gather_ms": {"p50": float(np.percentile(gather_ms, 50)), "max": float(np.max(gather_ms))} if gather_ms else None,
                      "chains_stitched": int(last["info"][0].item()) if last is not None else None,
                      **identities_carried(par, last, world),
                      "env": {k: os.environ.get(k) for k in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS", "MVMC_COMM_PRIORITY")},
                      "note": "pack -> ONE all-gather of fixed-size messages (live tracklets, f32) -> stitch, on a high-priority "
                              "communication stream behind an event; gather_ms = events around the collective on that stream"}
    stage_ms = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in ev.items() if v}
    if serial_ms is not None:
        stage_ms["one_step_alone"] = serial_ms
    if tail_events:
        te = np.array([[a.elapsed_time(b) for a, b in zip(e, e[1:])] for e in tail_events])
        stage_ms["tail_pack_gather_stitch"] = [float(x) for x in te.mean(axis=0)]
    if kern_events:
        # the kernel alone (events right around its launch): the timed region's launches, and every launch of the process -- the
        # average a profiler reports also contains the warm-up launches and the stand-alone reference launch
        stage_ms["chain_kernel_timed_region"] = float(np.mean([a.elapsed_time(b) for tm, a, b in kern_events if tm]))
        stage_ms["chain_kernel_all_launches"] = float(np.mean([a.elapsed_time(b) for tm, a, b in kern_events]))
    chain = with_ik and L > 1
    fused = chain and args.path == "fused"
    if chain and not fused:
        ik_launch_ms = [a.elapsed_time(b) for a, b in ik_events]
        als_launch_ms = [a.elapsed_time(b) for a, b in als_events]
        stage_ms["ik"] = float(np.sum(ik_launch_ms) / args.steps)     # all IK launches of one step
        stage_ms["als_temporal"] = float(np.sum(als_launch_ms) / args.steps)   # ALS on the match_spatial_time graphs
        stage_ms["other"] = stage_ms["total"] - stage_ms["ik"] - stage_ms["als_temporal"]
    if fused and out.get("phase") is not None:
        # one launch: the phases interleave across chains, so only per-chain shares of shader cycles can be given
        pc = out["phase"].cpu().numpy()
        tot = pc[:, 6].sum()
        stage_ms["chain_cycle_shares"] = {n: float(pc[:, k].sum() / tot) for k, n in
                                          enumerate(("graph", "als", "assign", "ik", "commit", "outputs"))}
        stage_ms["chain_mcycles_mean_max"] = [float(pc[:, 6].mean() / 1e6), float(pc[:, 6].max() / 1e6)]
    tracker_events = None
    if fused and "next_id" in out:
        # what the tracker did in one step of this rank (the reference's MvTracker: ids handed out = births, tracklets deleted = deaths)
        cnt = out["n_tracks"].cpu().numpy().reshape(-1, L)
        tracker_events = {"births": int(out["next_id"].sum().item()), "deaths": int(out["n_dead"].sum().item()),
                          "frames_where_the_count_changes": int((np.diff(cnt, axis=1) != 0).sum()),
                          "mean_live_tracklets": float(cnt.mean()),
                          # chains per step whose void word was set (more live tracklets than the chain kernel's table, a graph beyond
                          # its association variant) and that the repair tier re-ran inside the timed region; after it no word may
                          # be left (a step with one is void and ends the run, see finish())
                          "chains_repaired_per_step": float(np.mean(repaired_per_step)) if repaired_per_step else 0.0,
                          "capacity_word": int(out["void"].max().item()) if "void" in out else 0}
    if rank == 0:
        frames_total = F * world * args.steps
        value = frames_total / dt
        bpf = BYTES_PER_FRAME(C, Pn)
        extra_roof = {}
        dom = "ik" if with_ik else "assoc"
        dom_kernel = "ik1_kernel" if with_ik else "als4_kernel<float, 24>"
        if fused:
            # the whole step is ONE launch of chain_kernel over the rank's F frames
            dom, dom_kernel = "chain", "chain_kernel"
            launch_ms = stage_ms.get("chain_kernel_timed_region", stage_ms["total"])
            achieved = bpf * F / (launch_ms * 1e-3) / 1e9
        elif chain:
            # dominant kernel = the one with the larger share of the step; both are launched once per time step over all
            # chains, so one launch serves F/L frames
            if stage_ms["als_temporal"] > stage_ms["ik"]:
                dom, dom_kernel = "als", "als4_kernel<double, 32>"
                launch_ms = float(np.mean([x for i, x in enumerate(als_launch_ms) if i % L]))  # heads have no graph yet
            else:
                launch_ms = float(np.mean(ik_launch_ms))
            achieved = bpf * (F // L) / (launch_ms * 1e-3) / 1e9
        elif args.workload == "dlt":
            # config 2, the bytes the pass MOVES per frame: 12 C P 25 read (the OpenPose rows: x, y, score as float32) + one point of
            # {x, y, z, score} per person and per COCO-17 joint written, 16 B as float32 / 32 B as float64.  SURVEY.md 8(d) prices the
            # output at 25 joints (1,900 B at C5 P1); the path triangulates the 17 joints it uses (pose_def.py:262-270), so that figure
            # flattered `frac` by 7 % -- kept beside it as bytes_per_frame_survey_8d.  The DLT kernel dominates.
            dom, dom_kernel = "tri", "ingest_dlt3_kernel" if args.path == "fused" else "dlt_kernel"
            out_b = 16 if args.dlt_out == "f32" else 32
            bpf = 12 * C * Pn * 25 + out_b * Pn * 17
            extra_roof = {"bytes_per_frame_survey_8d": 12 * C * Pn * 25 + 16 * Pn * 25,
                          "bytes_per_frame_is": f"12 C P 25 read + {out_b} P 17 written (the COCO-17 joints, {args.dlt_out} points)"}
            launch_ms = stage_ms["tri"]
            achieved = bpf * F / (launch_ms * 1e-3) / 1e9
        else:
            launch_ms = stage_ms[dom]
            achieved = bpf * F / (launch_ms * 1e-3) / 1e9
        info = out.get("info")
        extra = {}
        if with_ik and info is not None:
            inf = info.reshape(-1, 8)
            ok = ~torch.isnan(inf[:, 1])
            extra = dict(ik_solves_per_step=int(ok.sum()), mean_nfev=float((inf[ok, 1] + inf[ok, 4]).mean()),
                         mean_njev=float(inf[ok, 6].mean()), eigensolver_fallbacks_per_solve=float(inf[ok, 7].mean()))
        # HBM-side bytes per launch come from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/aggregate_profiles.py)
        # of THIS command, recorded with a hash of the kernel sources: a record made from other sources is stale and reported as null
        traffic, traffic_note, fp64 = None, "no PMC record for this workload", None
        tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tp):
            rec = (json.load(open(tp)).get(f"{dom}:{F}x{C}x{Pn}") or {})
            if args.workload == "dlt" and args.dlt_out != "f32":
                traffic_note = "no PMC record for float64 points out (profiles/pmc_traffic.json holds the float32-output pass)"
            elif rec.get("src_sha") == kernel_sources_sha():
                traffic, traffic_note = rec.get("bytes"), f"profiles/{rec.get('source')}"
            elif rec:
                traffic_note = f"stale: profiles/{rec.get('source')} was measured on other kernel sources"
            # what bounds the path is the fp64 vector pipe, not HBM: wave instructions of the launch by class (SQ_INSTS_VALU_*_F64,
            # tools/prof_insts.sh + tools/aggregate_insts.py, same staleness rule) against this run's time per step
            mix = rec.get("inst_mix") or {}
            if mix.get("src_sha") == kernel_sources_sha():
                step_s = dt / args.steps
                tf = mix["flop_per_launch"] / step_s / 1e12
                fp64 = {"achieved": tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_VECTOR_PEAK_TFLOPS,
                        "f64_share_of_valu_insts": mix["f64_share_of_valu"], "lanes_active_per_valu_inst": mix["lanes_active_per_valu_inst"],
                        "valu_pipe_busy": mix["valu_busy_simd_cycles"] / (1024 * NOMINAL_CLOCK_HZ * step_s),
                        "lds_pipe_busy": mix["lds_busy_cu_cycles"] / (256 * NOMINAL_CLOCK_HZ * step_s),
                        **({"matrix_core_share_of_flop": mix["mfma_share_of_flop"], "matrix_core_insts_per_launch": mix["mfma_insts"]}
                           if "mfma_share_of_flop" in mix else {}),
                        "source": f"profiles/{mix.get('source')}",
                        "note": "flop = 64 lanes x (2 FMA + ADD + MUL + TRANS) fp64 wave instructions (+ 2,048 per v_mfma_f64_16x16x4_f64: same pipe, "
                                "same peak) of one launch (one step) / this run's "
                                "time per step; pipe shares = busy cycles of the launch's VALU (1024 SIMDs) and LDS (256 CUs) / the step at 2.4 GHz"}
        res = {
            "metric": "frames/s (assoc+triangulate+IK) at C=5,P=4,J=25" if with_ik else
                      ("frames/s (triangulate)" if args.workload == "dlt" else "frames/s (assoc+triangulate)"),
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "per_rank_ms_per_step": per_rank_ms, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"synthetic {F} frames/GPU, C={C}, P={Pn}, J=25: " + ("ingest + DLT of one cluster per person (config 2)"
                                    if args.workload == "dlt" else "affinity+ALS+DLT") +
                                   ((f"+IK, temporal chains of {L} frames (match_spatial_time + tracker; cold 50+50 nfev at the head, "
                                     f"warm 5+5 after), {('one launch per step, ' + (str(args.parts or L) + ' workgroup(s) per chain')) if args.path == 'fused' else 'one launch per stage'}" + (f", {args.overlap} steps in flight on alternating streams" if args.overlap > 1 else "") if L > 1 else "+IK, every frame cold-started (chain length 1, max_nfev 50+50)")
                                    if with_ik else ""),
                       "frames_per_gpu": F, "frames_total": F * world, "views": C, "people": Pn, "chain_len": L, "seed": args.seed,
                       "parallelism": f"frames x{world}" + (" (contiguous ranges of one scene)" if args.walk == "continuous" else " (a scene per rank)"),
                       "walk": args.walk,
                       "steps_in_flight": args.overlap, "hip_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "occlusion": args.occlusion, "spurious": args.spurious,
                       **({"tiled_from_frames": F_gen} if F_gen != F else {}),
                       **({"points_stored_as": args.dlt_out} if args.workload == "dlt" else {}), **extra},
            # what the measured library was built from (lib/BUILD_INFO.json; _cabi.load() has already refused a library whose
            # kernel-source hash is not this tree's); an MVMC_LIB_PATH library is an A/B partner and says so
            "build": (_cabi_build_info() or {"library": os.environ.get("MVMC_LIB_PATH"), "note": "MVMC_LIB_PATH: not the tree's build"}),
            "sustained": sustained,
            "host_io": host_io,
            "collective": collective,
            "als_iterations": als_histogram(out["als_it"].cpu().numpy()) if out.get("als_it") is not None else None,
            "tracker_events_per_step": tracker_events,
            "stages_ms": stage_ms,
            "roofline": {"bound": "hbm", "kernel": dom_kernel,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "launch_ms": launch_ms,
                         "traffic": traffic, "traffic_source": traffic_note, "bytes_per_frame": bpf, **extra_roof, "fp64": fp64,
                         "note": ("HBM-bound pass (one read of the keypoints, one store per point) beside fp64 issue; achieved = algorithmic "
                                  "bytes of the frames one launch serves / mean launch duration of the dominant kernel"
                                  if args.workload == "dlt" else
                                  "latency-bound path (dependent fp64 chains), not HBM bound (SURVEY.md F6); achieved = algorithmic "
                                  "bytes of the frames one launch serves / mean launch duration of the dominant kernel") +
                                 (f" ({args.overlap} launches share the GPU, so a launch lasts longer than ms_per_step)" if args.overlap > 1 else "") +
                                 "; see DESIGN.md"},
        }
        if chain and "gt_joints" in data:
            # accuracy of the last step's output against the generator's ground truth (the parity gates against the
            # reference are the tests'; this is the size-independent check at the benchmark's full size)
            n_t = out["n_tracks"][:F].cpu().numpy()
            jo = out["joints"][:F].cpu().numpy()
            gt = data["gt_joints"]
            errs = []
            for f in range(0, F, 13):
                if n_t[f] == Pn:
                    dmat = np.linalg.norm(jo[f, :Pn, None] - gt[f][None], axis=-1).mean(axis=-1)
                    errs.append(dmat.min(axis=1))
            errs = np.concatenate(errs) if errs else np.array([np.nan])
            res["accuracy"] = {"frames_with_all_people_tracked": float((n_t == Pn).mean()),
                               "joint_error_vs_ground_truth_cm": {"median": float(np.median(errs) * 100),
                                                                  "p95": float(np.quantile(errs, 0.95) * 100)},
                               "note": "synthetic ground truth (2 px keypoint noise); parity with the reference: tests/"}
            if sharded and args.walk == "continuous":
                try:
                    res["accuracy"]["stitch_vs_ground_truth_rank0"] = carries_against_ground_truth(out, data, L, Pn)
                except Exception as exc:      # (a diagnostic: never at the cost of the line)
                    res["accuracy"]["stitch_vs_ground_truth_rank0"] = {"error": repr(exc)[:200]}
        if args.cpu_frames > 0 and world == 1:
            if chain:
                workers = max(1, min(args.cpu_workers or min(16, os.cpu_count() or 1), F // L))
                try:
                    numpy_port = cpu_baseline_chain(data, L, workers)
                except Exception as exc:   # a worker pool that cannot start must not cost the benchmark line
                    print(f"cpu baseline: worker pool failed ({exc!r}); timing one chain in this process", file=sys.stderr)
                    numpy_port = cpu_baseline_chain(data, L, 1)
                try:
                    # the stronger of the two CPU restatements is the baseline; the NumPy / SciPy one (the reference-equivalent
                    # Python path) rides along
                    res["cpu_baseline"] = cpu_baseline_twin(data, L, F)
                    res["cpu_baseline"]["numpy_port"] = numpy_port
                except Exception as exc:
                    print(f"cpu baseline: C++ twin unavailable ({exc!r})", file=sys.stderr)
                    res["cpu_baseline"] = numpy_port
            else:
                res["cpu_baseline"] = cpu_baseline(data, args.cpu_frames, args.nfev_cold)
        return res
    return None


if __name__ == "__main__":
    main()
