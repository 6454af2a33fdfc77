"""N > 1 path on CPU: world_size-2 gloo run of the shard / all-gather / stitch logic (parallel.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from multiview_motion_capture_amd.parallel import gather_results, shard_range, stitch_identities


def test_shard_ranges_cover_and_balance():
    for n, w in ((10, 3), (200000, 8), (7, 8), (16, 2)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    F = 6
    out = dict(labels=torch.full((F, 20), rank, dtype=torch.int32),
               params=torch.arange(F * 4 * 68, dtype=torch.float64).reshape(F, 4, 68) + 1000 * rank,
               joints=torch.full((F, 4, 18, 3), float(rank), dtype=torch.float64))
    g = gather_results(out, world)
    ok = (g["labels"].shape == (world * F, 20) and g["params"].shape == (world * F, 4, 68)
          and all(bool((g["labels"][r * F:(r + 1) * F] == r).all()) for r in range(world))
          and all(bool((g["joints"][r * F:(r + 1) * F] == float(r)).all()) for r in range(world))
          and bool(torch.equal(g["params"][F:2 * F], torch.arange(F * 4 * 68, dtype=torch.float64).reshape(F, 4, 68) + 1000)))
    q.put((rank, ok))
    dist.destroy_process_group()


def test_gather_results_gloo_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_stitch_identities_across_a_shard_boundary():
    rng = np.random.default_rng(0)
    prev = rng.normal(size=(5, 18, 3))
    prev[3] = np.nan  # empty slot
    perm = [2, 0, 4, 1]
    nxt = np.full((5, 18, 3), np.nan)
    for k, src in enumerate(perm):
        nxt[k] = prev[src] + rng.normal(scale=0.01, size=(18, 3))
    pairs = stitch_identities(prev, nxt)
    assert sorted(pairs) == sorted((src, k) for k, src in enumerate(perm))
    far = nxt.copy()
    far[0] += 10.0
    assert (perm[0], 0) not in stitch_identities(prev, far)
