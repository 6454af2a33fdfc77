"""The long ALS runs of BASELINE config 5 against the oracle -- the regime the short fixtures do not reach.

A config-5 step (C8 P8, seed 20260104, 25,008 frames, chains of 16) has 1,563 chain heads; a head's spatial graph has 64 nodes and
rank 16, and ~12 % of them iterate 500 times or more, ~0.8 % of all graphs up to the cap of 1,000 (mv_association.py:263-312: the loop
ends at maxIter whatever the residuals).  It turns out that EVERY head of this workload iterates 500 times or more (p50 700) and 188 of
the 1,563 (12 %) reach the cap.  Below the cap the device must agree with oracle_np.match_als exactly in X_bin and labels on every graph,
and in the iteration count up to the one decision that is a rounding matter: the loop stops when two fp64 norms are both below 1e-4,
and a norm summed in another order can cross that line one iteration earlier or later (1 of 1,375 runs here: 831 against 832, same
X_bin) -- a difference of at most one iteration on at most 1 % of the runs is allowed, counted and printed.  AT the cap the result is
whatever iterate number 1,000 happens to be, which depends on the summation order of the BLAS behind NumPy's products
(profiles/r05_assoc_soak.txt): agreement there is counted, printed and recorded (profiles/r06_als_cap_gate.txt); observed: all 188."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT
from helpers import als_oracle_job, frame_nodes, oracle_ingest

pytestmark = pytest.mark.gpu
F, C, P, L, SEED = 25008, 8, 8, 16, 20260104


def test_every_long_als_run_of_a_config_5_step_equals_the_oracle():
    from multiview_motion_capture_amd import synth
    from multiview_motion_capture_amd.pipeline import HotPath
    data = synth.generate(F, C, P, SEED, chain_len=L)                     # the benchmark's own step (float32 keypoints)
    heads = np.arange(0, F, L)
    assert len(heads) == 1563
    d = torch.device("cuda:0")
    hp = HotPath(data["K"], data["Rt"], device=d)
    kps_h, cnt_h = data["kps25"][heads], data["counts"][heads]
    assoc = hp.associate(torch.from_numpy(kps_h).to(d), torch.from_numpy(cnt_h).to(d), want_mats=True)
    torch.cuda.synchronize()
    iters = assoc["iters"].cpu().numpy()
    long_runs = np.nonzero(iters >= 500)[0]
    assert len(long_runs) >= 100, f"only {len(long_runs)} graphs of 500 iterations or more: not the workload this gate is for"
    S, xb, lab = assoc["S"].cpu().numpy(), assoc["x_bin"].cpu().numpy(), assoc["labels"].cpu().numpy()
    k17_o, cnt_o = oracle_ingest(kps_h[long_runs].astype(np.float64), cnt_h[long_runs])
    # the oracle on the affinity the device built (bit-exact against the oracle's own: tests/test_gpu_assoc_dlt_fk.py,
    # test_gpu_config5_c8p8.py), eight spawned workers (NumPy only; the GPU stays with this process)
    jobs, sizes = [], []
    for r, h in enumerate(long_runs):
        pts, _, dim, _ = frame_nodes(k17_o[r], cnt_o[r])
        n = len(pts)
        jobs.append((S[h, :n, :n].copy(), dim))
        sizes.append(n)
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(8) as pool:
        done = pool.map(als_oracle_job, jobs, chunksize=16)
    rows = []
    for (xb_o, lab_o, it_o), n, h in zip(done, sizes, long_runs):
        same_x = bool(np.array_equal(xb[h, :n, :n].astype(bool), xb_o))
        same_l = bool(np.array_equal(lab[h, :n], lab_o))
        rows.append((int(heads[h]), int(iters[h]), int(it_o), same_x, same_l))
    capped = [r for r in rows if r[1] >= 1000 or r[2] >= 1000]
    free = [r for r in rows if not (r[1] >= 1000 or r[2] >= 1000)]
    bad_free = [r for r in free if not (r[3] and r[4])]
    off_by = [r for r in free if r[1] != r[2]]
    cap_iters = sum(r[1] == r[2] for r in capped)
    cap_x = sum(r[3] for r in capped)
    cap_l = sum(r[4] for r in capped)
    text = (f"config 5 step (C{C} P{P}, seed {SEED}, {F} frames): {len(heads)} chain heads, {len(rows)} spatial graphs of >= 500 ALS iterations\n"
            f"  below the cap: {len(free)} graphs -- X_bin and labels equal to oracle_np.match_als on {len(free) - len(bad_free)}, the iteration "
            f"count on {len(free) - len(off_by)} (the others: {[(r[1], r[2]) for r in off_by][:8]})\n"
            f"  at the cap of 1000: {len(capped)} graphs -- iteration count equal on {cap_iters}, X_bin on {cap_x}, labels on {cap_l}\n"
            f"  iterations of the long runs: p50 {int(np.median([r[1] for r in rows]))}, max {max(r[1] for r in rows)}\n")
    print("\n" + text)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir) and os.access(out_dir, os.W_OK):
        with open(os.path.join(out_dir, "als_cap_gate.txt"), "w") as f:
            f.write(text)
            f.write(json.dumps({"differing_capped_frames": [r[0] for r in capped if not (r[3] and r[4])]}) + "\n")
    assert not bad_free, f"ALS runs below the cap differ from the oracle: {bad_free[:5]}"
    assert len(off_by) <= max(1, len(free) // 100) and all(abs(r[1] - r[2]) <= 1 for r in off_by), off_by[:8]
    assert cap_iters == len(capped), "a run that reaches the cap on one side must reach it on the other"
    # (the labels of a capped run depend on the BLAS's summation order on the oracle's side; the soak of round 5 saw 2 of 960 frames
    # differ.  A majority is demanded so that a systematic difference cannot hide here.)
    assert cap_l >= 0.8 * len(capped), (cap_l, len(capped))
