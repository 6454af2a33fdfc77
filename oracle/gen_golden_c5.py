"""Golden vectors of the REFERENCE tracker on BASELINE config 5's geometry (C = 8 views, P = 8 people, seed 20260104).

TEST INFRASTRUCTURE (build container only).  Runs the reference's own MvTracker.update_4d (motion_capture.py:873-963) from
/root/reference/src through ``oracle/ref_shim.py`` over the first 4 chains x 16 frames of the synthetic C8 P8 workload, one fresh
tracker per chain (the benchmark's protocol); only inputs' checksum and outputs are written: tests/golden/synth_c5_tracker.npz
(the tracklet table after every frame: ids, states, hits, lengths, parameters, joints; per-solve view counts and SciPy status).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_c5.py

Single-threaded BLAS, as oracle/gen_golden_ikconv.py (LAPACK's rounding depends on the thread count)."""
import os
import sys

os.environ["OPENBLAS_NUM_THREADS"] = "1"
os.environ["OMP_NUM_THREADS"] = "1"

import numpy as np  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_shim  # noqa: E402
import gen_golden_ikconv as gk  # noqa: E402

C5 = dict(n_frames=64, n_views=8, n_people=8, seed=20260104, chain_len=16)


def main():
    m = ref_shim.load_modules()
    fix, cases = gk.run_synth_tracker(m, C5, T=16)
    fix["solve_views"] = np.array([len(c["poses"]) for c in cases])
    np.savez_compressed(os.path.join(gk.OUT, "synth_c5_tracker.npz"), **fix)
    print("saved synth_c5_tracker.npz:", int(fix["n_solves"].sum()), "solves over", C5["n_frames"], "frames; tracklets per frame",
          np.bincount(fix["n_tracks"]), "views per solve", np.bincount(fix["solve_views"]))


if __name__ == "__main__":
    main()
