"""N > 1 path on the CPU: world-size-2 and -3 gloo runs of parallel.run_sharded end to end -- shard ranges, message packing, the one
all-gather, and the stitch of identities across chain and shard boundaries -- with the compute and the device kernels replaced by
host stand-ins that speak the same message format (oracle/stitch_np.py; there is no GPU here).  The device kernels themselves are
checked against the same host restatement in tests/test_gpu_parallel.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import stitch_np as sn
from multiview_motion_capture_amd.parallel import chains_cap, run_sharded, shard_range, unpack_message

L, T, P = 4, 8, 3
N_CHAINS = 7          # 28 frames: shards of 4 and 3 chains -- unequal on purpose


def test_shard_ranges_cover_and_balance():
    for n, w in ((10, 3), (200000, 8), (7, 8), (16, 2), (12500, 8)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1 and max(sizes) == chains_cap(n, w)


def fake_sequence(seed=0):
    """Tracker tables of a continuous scene processed in chains of L frames: P people walk smoothly; inside every chain the slots
    hold them in a chain-specific order with chain-local ids; person 2 is absent in chains 3-4 (dies, is re-born later).
    -> per-frame arrays as mvmc_chain_run writes them + the person behind every (frame, slot)."""
    rng = np.random.default_rng(seed)
    F = N_CHAINS * L
    base = rng.normal(size=(P, 18, 3)) * 0.2 + rng.uniform(-2, 2, size=(P, 1, 3))
    walk = np.cumsum(rng.normal(scale=0.01, size=(F, P, 1, 3)), axis=0)
    gt = base[None] + walk
    params = np.zeros((F, T, 68))
    joints = np.zeros((F, T, 18, 3))
    meta = np.zeros((F, T, 4), dtype=np.int32)
    n_tracks = np.zeros(F, dtype=np.int32)
    next_id = np.zeros(N_CHAINS, dtype=np.int32)
    who = -np.ones((F, T), dtype=np.int32)
    for b in range(N_CHAINS):
        people = [p for p in range(P) if not (p == 2 and b in (3, 4))]
        order = rng.permutation(people)
        next_id[b] = len(order)
        for t in range(L):
            f = b * L + t
            n_tracks[f] = len(order)
            for s, p in enumerate(order):
                joints[f, s] = gt[f, p] + rng.normal(scale=0.002, size=(18, 3))
                params[f, s] = rng.normal(size=68)
                meta[f, s] = (s, 2, t + 1, t + 1)
                who[f, s] = p
            joints[f, len(order):] = 7.0   # stale rows beyond n_tracks must be ignored (the table is not NaN-padded)
    return dict(params=params, joints=joints, meta=meta, n_tracks=n_tracks, next_id=next_id, who=who)


def _pack(out, next_id, chain_len, b_cap, row_cap, t_msg=None, void_words=None):
    return torch.from_numpy(sn.pack_np(out["params"].numpy(), out["joints"].numpy(), out["meta"].numpy(), out["n_tracks"].numpy(),
                                       next_id.numpy(), chain_len, b_cap, row_cap, t_msg=t_msg,
                                       void_words=None if void_words is None else void_words.numpy()))


def _stitch(msgs, b_cap, t_max, row_cap, max_dist):
    return {k: torch.from_numpy(v) for k, v in sn.stitch_np(msgs.numpy(), b_cap, t_max, row_cap, max_dist).items()}


def _check_identities(gid, seq, b_cap, world):
    """every (chain, local id) -> global id; the same person must keep one global id while present in consecutive chains"""
    owner = {}
    spans = [shard_range(N_CHAINS, r, world) for r in range(world)]
    # row of chain g in the gid table: chains of rank r start at the running count of the earlier ranks
    for g in range(N_CHAINS):
        f = g * L
        for s in range(seq["n_tracks"][f]):
            person, lid = seq["who"][f, s], seq["meta"][f, s, 0]
            owner.setdefault(person, []).append((g, int(gid[g, lid])))
    for person, lst in owner.items():
        for (g0, a), (g1, b) in zip(lst, lst[1:]):
            if g1 == g0 + 1:
                assert a == b, (person, g0, g1, a, b)
            else:
                assert a != b   # the person was away for whole chains: a new identity
    all_ids = sorted({i for lst in owner.values() for _, i in lst})
    assert all_ids == list(range(len(all_ids)))
    return len(all_ids)


def _worker(rank, world, port, q, mode="plain"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seq = fake_sequence()
    lo, hi = shard_range(N_CHAINS, rank, world)
    # mode "one_rank_widened": rank 1's tables have 16 slots per frame (what tracker.repair_chains leaves behind when a repaired chain
    # needs more than 8), rank 0's have 8 -- both must send messages of the same size and layout (t_msg)
    widen = mode == "one_rank_widened" and rank == 1
    t_msg = 16 if mode == "one_rank_widened" else None
    t_view = t_msg or T

    def compute():
        sl = slice(lo * L, hi * L)
        tabs = {k: seq[k][sl] for k in ("params", "joints", "meta")}
        if widen:
            for k, v in tabs.items():
                w = np.zeros((v.shape[0], 16) + v.shape[2:], dtype=v.dtype)
                w[:, :T] = v
                tabs[k] = w
        out = {k: torch.from_numpy(v) for k, v in tabs.items()}
        out.update(n_tracks=torch.from_numpy(seq["n_tracks"][sl]), next_id=torch.from_numpy(seq["next_id"][lo:hi]))
        if mode == "void_rank_1":
            out["void_words"] = torch.tensor([0, 0, 1 if rank == 1 else 0], dtype=torch.int32)
        return out

    res = run_sharded(compute, L, N_CHAINS, rank, world, rows_per_frame=P, pack=_pack, stitch=_stitch, t_msg=t_msg)
    info = res["info"].numpy()
    if mode == "void_rank_1":
        # every rank sees rank 1's void word in the gathered messages: the step is void everywhere, nobody read a device word
        from multiview_motion_capture_amd.parallel import check_stitch_info
        try:
            check_stitch_info(res)
            raised = False
        except RuntimeError as exc:
            raised = "void" in str(exc)
        q.put((rank, bool(raised and info[2] & 4), 0, 0, 0, b""))
        dist.destroy_process_group()
        return
    ok = info[0] == N_CHAINS and info[2] == 0
    n_ids = _check_identities(res["gid"].numpy(), seq, res["b_cap"], world)
    ok = ok and all(res["messages"][r].numel() == res["messages"][0].numel() for r in range(world))
    # every shard's rows arrived: frames, slots and joints of rank r are in message r
    got = 0
    for r in range(world):
        u = unpack_message(res["messages"][r].numpy(), res["b_cap"], t_view, res["row_cap"])
        l2, h2 = shard_range(N_CHAINS, r, world)
        sl = slice(l2 * L, h2 * L)
        ok = ok and u["n_chains"] == h2 - l2 and u["n_rows"] == int(seq["n_tracks"][sl].sum()) == u["rows_wanted"]
        fr, slot = u["row_meta"][:, 0], u["row_meta"][:, 1]
        ok = ok and np.array_equal(u["row_joints"], seq["joints"][sl][fr, slot].astype(np.float32))
        ok = ok and np.array_equal(u["row_params"], seq["params"][sl][fr, slot].astype(np.float32))
        got += u["n_rows"]
    q.put((rank, bool(ok), int(n_ids), int(info[3]), got, res["gid"].numpy().tobytes()))
    dist.destroy_process_group()


def _run_world(world, mode):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    return res


def test_a_void_step_of_one_rank_is_seen_by_every_rank():
    res = _run_world(2, "void_rank_1")
    assert [r[0] for r in res] == [0, 1] and all(r[1] for r in res)


@pytest.mark.parametrize("world,mode", [(2, "plain"), (3, "plain"), (2, "one_rank_widened")])  # 7 chains: shards of 4 + 3, of 3 + 2 + 2
def test_run_sharded_gloo_end_to_end(world, mode):
    res = _run_world(world, mode)
    assert [r[0] for r in res] == list(range(world)) and all(r[1] for r in res)
    assert all(r[5] == res[0][5] for r in res)         # every rank holds the same global identities
    assert all(r[2] == 4 for r in res)                 # 3 people + the re-born one
    seq = fake_sequence()
    assert res[0][4] == int(seq["n_tracks"].sum())
    # the same sequence stitched in one process (world 1) gives the same identities: sharding does not change the result
    one = sn.stitch_np(sn.pack_np(seq["params"], seq["joints"], seq["meta"], seq["n_tracks"], seq["next_id"], L, N_CHAINS,
                                  N_CHAINS * L * P)[None], N_CHAINS, T, N_CHAINS * L * P)
    two = np.frombuffer(res[0][5], dtype=np.int32).reshape(-1, 16)
    # (with one rank's tables widened to 16 slots the identities are the same again: the message's slot count is fixed, not the tables')
    # (gid rows are global chain indices, whatever the sharding)
    assert np.array_equal(one["gid"][:N_CHAINS], two[:N_CHAINS])
    assert one["info"][3] == res[0][3]


def test_stitch_rule_on_one_boundary():
    rng = np.random.default_rng(0)
    prev = rng.normal(size=(4, 18, 3)).astype(np.float32)
    perm = [2, 0, 3, 1]
    nxt = np.stack([prev[src] + rng.normal(scale=0.01, size=(18, 3)).astype(np.float32) for src in perm])
    assert sorted(sn.match_boundary(prev, nxt, 0.5)) == sorted((src, k) for k, src in enumerate(perm))
    far = nxt.copy()
    far[0] += 10.0
    assert (perm[0], 0) not in sn.match_boundary(prev, far, 0.5)
    assert sn.match_boundary(prev[:0], nxt, 0.5) == []
