"""bench.py host logic that needs no GPU: the multi-rank launcher must not leave ranks behind when one of them dies."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_a_dead_rank_ends_the_whole_job_quickly():
    # `python bench.py --gpus 2` starts two child ranks.  Here there is no GPU (and on a one-GPU box rank 1 has no device): a rank fails at
    # torch.cuda.set_device.  The launcher has to notice, stop the other rank -- which would otherwise wait in the rendezvous for its
    # time-out -- and return a non-zero code.
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--frames", "64", "--steps", "1",
                        "--warmup", "0", "--cpu-frames", "0", "--sustain", "0"], capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert r.returncode != 0
    assert "the other ranks were stopped" in r.stderr or "exited with code" in r.stderr
    assert r.stdout.strip() == ""            # no bench line from a failed job
    assert time.time() - t0 < 200


def test_kernel_source_hash_is_stable_and_sensitive():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.kernel_sources_sha()
    assert a == bench.kernel_sources_sha() and len(a) == 16


def test_als_histogram_and_the_defaults_of_the_command_line():
    """The ALS iteration histogram of a bench line and the flags added in round 5."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    h = bench.als_histogram(np.array([[44, 140, 0], [205, 1000, 999]]))
    assert h["graphs"] == 5 and h["min"] == 44 and h["max"] == 1000 and h["share_at_the_1000_cap"] == 0.2
    assert sum(h["histogram"].values()) == 5 and h["histogram"]["1000"] == 1 and h["histogram"]["[500, 1000)"] == 1
    assert bench.als_histogram(np.zeros(4, dtype=np.int32)) is None
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    for flag in ("--force-collective", "--dlt-out", "--other-configs", "--frames-total", "--walk"):
        assert flag in r.stdout
    assert "--big-split" not in r.stdout                                    # (retired in round 6)
    # the strong-scaling form refuses a total that does not divide into whole chains per rank
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames-total", "1000"], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode != 0 and "--frames-total" in r.stderr
    assert bench.OTHER_CONFIGS[-1][1][-2:] == ["--dlt-out", "f64"]           # config 2 with the reference's output dtype rides along
    assert bench.DEFAULT_NCCL_MAX_NCHANNELS == "4" and bench.OTHER_CONFIGS[1][1][:6] == ["--views", "8", "--people", "8", "--frames", "25008"]
