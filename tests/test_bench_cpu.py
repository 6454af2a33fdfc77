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
