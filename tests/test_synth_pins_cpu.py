"""CPU: pins of the synthetic generator (the build's own; SURVEY.md section 8c last bullet) and the oracle tracker on the benchmark's
workload: seed -> checksum of the generated tensors, and oracle/tracker_np.py == the reference's own run (tests/golden/
synth_c4_tracker.npz, written by oracle/gen_golden_ikconv.py) on the first chain of the 64-frame subset of synthetic config 4."""
import numpy as np

import oracle_np as o
import tracker_np as tk
from conftest import load_golden
from helpers import oracle_ingest

CHECKSUMS = {  # (n_frames, views, people, seed, chain_len) -> sum |kps25| in float64
    (64, 5, 4, 20260103, 16): None,   # filled from the fixture below (the value the reference run saw)
}


def test_generator_is_pinned_by_seed():
    from multiview_motion_capture_amd import synth
    g = load_golden("synth_c4_tracker.npz")
    a = synth.generate(64, 5, 4, 20260103, chain_len=16)
    b = synth.generate(64, 5, 4, 20260103, chain_len=16)
    assert np.array_equal(a["kps25"], b["kps25"]) and np.array_equal(a["gt_order"], b["gt_order"])
    assert float(np.abs(a["kps25"].astype(np.float64)).sum()) == float(g["kps25_checksum"])
    assert a["kps25"].dtype == np.float32 and a["kps25"].shape == (64, 5, 4, 25, 3) and (a["counts"] == 4).all()
    c = synth.generate(64, 5, 4, 20260104, chain_len=16)
    assert not np.array_equal(a["kps25"], c["kps25"])
    # a longer sequence with the same seed starts with the same cameras; frame_seed decouples frames from cameras (rank shards)
    e = synth.generate(32, 5, 4, 20260103, chain_len=16, frame_seed=5)
    assert np.array_equal(e["K"], a["K"]) and np.array_equal(e["Rt"], a["Rt"]) and not np.array_equal(e["kps25"], a["kps25"][:32])


def test_oracle_tracker_equals_the_reference_on_the_synthetic_workload():
    from multiview_motion_capture_amd import synth
    g = load_golden("synth_c4_tracker.npz")
    L = int(g["chain_len"])
    data = synth.generate(64, 5, 4, 20260103, chain_len=16)
    k17, cnt = oracle_ingest(data["kps25"][:L].astype(np.float64), data["counts"][:L])
    tr = tk.OracleTracker(data["K"], data["Rt"], data["P"])
    # the fixture was generated with single-threaded BLAS (OPENBLAS_NUM_THREADS=1, see oracle/gen_golden_ikconv.py); LAPACK's
    # rounding -- and with it the truncated solves -- depends on the thread count, so the comparison runs the same way
    from threadpoolctl import threadpool_limits
    worst = 0.0
    with threadpool_limits(limits=1):
        for f in range(L):
            tr.update(f + 1, [[k17[f, c, p] for p in range(cnt[f, c])] for c in range(5)])
            got = [(t.tid, t.state, t.hits, t.length) for t in tr.tracklets]
            exp = [tuple(int(v) for v in r) for r in g["meta"][f] if r[0] >= 0]
            assert got == exp, (f, got, exp)
            for s, t in enumerate(tr.tracklets):
                worst = max(worst, np.abs(t.joints - g["joints"][f, s]).max())
    print("oracle tracker vs the reference's run, first chain: max joint difference", worst)
    assert worst == 0.0   # bit-exact
    assert len(tr.solves) == int(g["n_solves"][:L].sum())


def test_oracle_tracker_equals_the_reference_on_config_5_geometry():
    """The same pin at C8 P8 (tests/golden/synth_c5_tracker.npz, oracle/gen_golden_c5.py: the reference's MvTracker.update_4d on the
    first chains of synthetic config 5, seed 20260104): 64-node match_spatial graphs, 72-node match_spatial_time graphs, eight solves per
    frame.  First six frames of chain 0 (the oracle needs ~1.5 s per frame here), bit for bit."""
    from multiview_motion_capture_amd import synth
    from threadpoolctl import threadpool_limits
    g = load_golden("synth_c5_tracker.npz")
    F, L, C, P = int(g["n_frames"]), int(g["chain_len"]), int(g["n_views"]), int(g["n_people"])
    assert (C, P, int(g["seed"])) == (8, 8, 20260104)
    data = synth.generate(F, C, P, int(g["seed"]), chain_len=L)
    assert float(np.abs(data["kps25"].astype(np.float64)).sum()) == float(g["kps25_checksum"])
    n = 6
    k17, cnt = oracle_ingest(data["kps25"][:n].astype(np.float64), data["counts"][:n])
    tr = tk.OracleTracker(data["K"], data["Rt"], data["P"])
    with threadpool_limits(limits=1):
        for f in range(n):
            tr.update(f + 1, [[k17[f, c, p] for p in range(cnt[f, c])] for c in range(C)])
            got = [(t.tid, t.state, t.hits, t.length) for t in tr.tracklets]
            exp = [tuple(int(v) for v in r) for r in g["meta"][f] if r[0] >= 0]
            assert got == exp, (f, got, exp)
            for s, t in enumerate(tr.tracklets):
                assert np.array_equal(t.joints, g["joints"][f, s]), (f, s)
    assert len(tr.solves) == int(g["n_solves"][:n].sum())
