"""A small run of tools/oracle_soak.py inside the suite: the chain kernel against the deterministic oracle tracker on seeds the other tests
do not use -- clean, and with occlusion + false detections (births, deaths, two-view clusters of false detections).  The full net
(18 seeds x 5 workloads, profiles/r05_oracle_soak.txt) is what found the two differences fixed in round 5 (the post-optimisation of
two-view clusters; NumPy's float32 exp in the affinity): this keeps both shut."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_three_seeds_of_configs_4_and_5_against_the_oracle_tracker():
    import oracle_soak as soak
    res = soak.run([soak.WORKLOADS[0], soak.WORKLOADS[2], soak.WORKLOADS[3]], [21, 22, 23], workers=min(14, os.cpu_count() or 4))
    for r in res:
        C, P, n_chains, occ, spur = r["workload"]
        assert r["frames"] == 3 * n_chains * 16
        assert r["tables_equal"] == r["frames"], r          # ids, states, hits, lengths on EVERY frame
        # the frame's ALS iteration count: equal wherever the run converges (S is bit-exact); a run at its cap of 1000 is an unconverged
        # iteration whose bits depend on the summation order of the factor products, and the frames behind it inherit its tracklets
        assert r["als_equal"] >= r["frames"] - 16 * r["als_capped"], r     # (observed: equal on every frame, capped ones included)
        if occ == 0.0:
            assert r["worst"] < 1e-6, r                      # observed: 1e-13 .. 2e-9 m
        else:
            # births from two false detections (hits = 1, gone on the next frame) are the only tracklets that may differ: a kept
            # post-optimisation trial of metres (finite differences there, analytic here) in front of a chaotic 50 + 50 solve
            assert all(hits <= 2 for _, hits in r["above_1e6"]) and len(r["above_1e6"]) <= 0.005 * r["tracklet_frames"], r
