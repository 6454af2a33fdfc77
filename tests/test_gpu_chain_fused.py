"""The persistent chain kernel (mvmc_chain_run: one workgroup per chain, one launch per shard) against the launch-per-stage
path (ChainTracker.step: mvmc_affinity / mvmc_st_affinity / mvmc_als_associate / mvmc_track_assign / mvmc_ik_solve /
mvmc_track_commit per time step).  Same device code, different scheduling: every output must be bit-identical."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(B, L, seed=20260103, people=4, parts=None):
    from multiview_motion_capture_amd import synth
    from multiview_motion_capture_amd.pipeline import HotPath
    from multiview_motion_capture_amd.tracker import run_chains, run_chains_fused
    data = synth.generate(B * L, 5, people, seed, chain_len=L)
    hp = HotPath(data["K"], data["Rt"])
    kps = torch.from_numpy(data["kps25"]).cuda()
    cnt = torch.from_numpy(data["counts"]).cuda()
    a = run_chains(hp, kps, cnt, L, want_info=True)
    b = run_chains_fused(hp, kps, cnt, L, want_info=True, parts=parts)
    torch.cuda.synchronize()
    return ({k: v.cpu().numpy() for k, v in a.items()},
            {k: v.cpu().numpy() for k, v in b.items() if isinstance(v, torch.Tensor)})


@pytest.mark.parametrize("B,L,people,parts", [(48, 8, 4, None), (7, 16, 3, 1), (300, 4, 4, 2)])
def test_fused_equals_staged_bit_for_bit(B, L, people, parts):
    """parts = workgroups per chain: None = one per frame (the default), 1 = one persistent workgroup per chain."""
    a, b = _both(B, L, people=people, parts=parts)
    if parts != 1:
        assert int(b["flags"][B]) == 0, "a hand-over between the workgroups of a chain timed out"
        assert (b["flags"][:B] == (parts or L)).all()
    for k in ("n_tracks", "meta", "n_dead"):
        assert np.array_equal(a[k], b[k]), k
    for k in ("params", "joints", "ik_info"):
        assert a[k].shape == b[k].shape
        assert np.array_equal(np.isnan(a[k]), np.isnan(b[k])), k
        m = ~np.isnan(a[k])
        assert np.array_equal(a[k][m], b[k][m]), k
    assert (a["n_tracks"] > 0).all()
    # the kernel reports where each chain spent its cycles and how many ALS iterations its graphs took
    pc = b["phase_cycles"]
    assert pc.shape == (B, 8) and (pc[:, :6] > 0).all() and (pc[:, 6] >= pc[:, :6].sum(1) * 0.99).all()
    assert (pc[:, 7] == (parts or L)).all()
    assert b["als_iters"].shape == (B, L) and (b["als_iters"] > 0).all()


@pytest.mark.parametrize("people,views,force_big", [(4, 5, False), (4, 5, True), (8, 8, False)])
def test_ready_queue_hand_over_is_bit_identical_to_the_static_mapping(people, views, force_big):
    """hand_over = "queue": a workgroup draws a ticket when it starts and takes the chain that has been ready longest -- nothing
    depends on the order in which workgroups are dispatched (what the multi-GPU path uses).  More workgroups than slots (900 chains of 8
    frames on the SMALL layout: 7,200 workgroups for 768 slots), both layouts: same results as the static mapping bit for bit, every
    chain's flag at its last part, every ring entry filled, no time-out."""
    from multiview_motion_capture_amd import synth
    from multiview_motion_capture_amd.pipeline import HotPath
    from multiview_motion_capture_amd.tracker import check_chain_flags, run_chains_fused
    B, L = (900, 8) if views == 5 else (300, 4)
    data = synth.generate(B * L, views, people, 20260111, chain_len=L)
    hp = HotPath(data["K"], data["Rt"])
    kps, cnt = torch.from_numpy(data["kps25"]).cuda(), torch.from_numpy(data["counts"]).cuda()
    ref = run_chains_fused(hp, kps, cnt, L, hand_over="static", force_big=force_big)
    for parts in (None, 2):
        q = run_chains_fused(hp, kps, cnt, L, hand_over="queue", force_big=force_big, parts=parts)
        torch.cuda.synchronize()
        check_chain_flags(q)
        np_ = parts or L
        fl = q["flags"].cpu().numpy()
        assert (fl[:B] == np_).all() and fl[B] == 0
        assert fl[2 * B + 4] == B * np_ and fl[2 * B + 5] == B * (np_ - 1)          # tickets drawn, ring entries written
        ring = fl[2 * B + 6:2 * B + 6 + B * (np_ - 1)].astype(np.int64) - 1
        assert (ring >= 0).all() and len(set(ring.tolist())) == B * (np_ - 1)       # every (chain, later part) exactly once
        for k in ("meta", "n_tracks", "n_dead", "next_id"):
            assert torch.equal(q[k], ref[k]), k
        for k in ("params", "joints"):
            assert torch.equal(torch.nan_to_num(q[k]), torch.nan_to_num(ref[k])), k
        # the ticket protocol (the default): the static mapping indexed by a ticket drawn at start -- same results, every ticket drawn
        tk = run_chains_fused(hp, kps, cnt, L, hand_over="ticket", force_big=force_big, parts=parts)
        torch.cuda.synchronize()
        check_chain_flags(tk)
        fl = tk["flags"].cpu().numpy()
        assert (fl[:B] == np_).all() and fl[B] == 0 and fl[2 * B + 4] == B * np_
        for k in ("meta", "n_tracks", "n_dead", "next_id"):
            assert torch.equal(tk[k], ref[k]), k
        for k in ("params", "joints"):
            assert torch.equal(torch.nan_to_num(tk[k]), torch.nan_to_num(ref[k])), k
    dflt = run_chains_fused(hp, kps, cnt, L, force_big=force_big)
    torch.cuda.synchronize()
    assert int(dflt["flags"][2 * B + 4]) == B * L            # the default protocol draws tickets
    with pytest.raises(ValueError):
        run_chains_fused(hp, kps, cnt, L, hand_over="fifo")


def test_sizes_outside_the_arena_are_refused():
    from multiview_motion_capture_amd import synth
    from multiview_motion_capture_amd._cabi import MvmcError
    from multiview_motion_capture_amd.pipeline import HotPath
    from multiview_motion_capture_amd.tracker import run_chains_fused
    data = synth.generate(8, 12, 8, 20260104, chain_len=4)     # 96 nodes per frame: beyond both layouts (config 5's 64 fit the BIG one)
    hp = HotPath(data["K"], data["Rt"])
    with pytest.raises(MvmcError):
        run_chains_fused(hp, torch.from_numpy(data["kps25"]).cuda(), torch.from_numpy(data["counts"]).cuda(), 4)
    with pytest.raises(ValueError):
        run_chains_fused(hp, torch.from_numpy(data["kps25"]).cuda(), torch.from_numpy(data["counts"]).cuda(), 3)
    with pytest.raises(ValueError):   # parts must divide the chain length
        run_chains_fused(hp, torch.from_numpy(data["kps25"]).cuda(), torch.from_numpy(data["counts"]).cuda(), 4, parts=3)


def test_graph_larger_than_the_kernels_als_variant_is_reported():
    """Padded sizes up to 40 nodes are accepted, but a frame whose actual graph exceeds the workgroup ALS variant (24 nodes
    without tracklets) must raise the launch's error word -- never a silently empty result."""
    from multiview_motion_capture_amd import synth
    from multiview_motion_capture_amd.pipeline import HotPath
    from multiview_motion_capture_amd.tracker import check_chain_flags, run_chains_fused
    data = synth.generate(2 * 4, 5, 8, 20260105, chain_len=4)          # 5 views x 8 people = 40 nodes per frame
    hp = HotPath(data["K"], data["Rt"])
    kps, cnt = torch.from_numpy(data["kps25"]).cuda(), torch.from_numpy(data["counts"]).cuda()
    out = run_chains_fused(hp, kps, cnt, 4)
    with pytest.raises(ValueError):
        check_chain_flags(out)
    assert out["void"].cpu().tolist() == [4, 4]
    # ... and the repair tier runs those chains through the per-stage entry points (generic association variant, 80 nodes)
    from multiview_motion_capture_amd.tracker import repair_chains, run_chains
    assert repair_chains(hp, kps, cnt, out) == 2
    check_chain_flags(out)
    ref = run_chains(hp, kps, cnt, 4)
    check_chain_flags(ref)
    assert torch.equal(out["n_tracks"], ref["n_tracks"]) and int(ref["n_tracks"].min()) == 8
    n = ref["n_tracks"].cpu().numpy()
    for f in range(len(n)):
        assert torch.equal(out["meta"][f, :n[f]], ref["meta"][f, :n[f]])
        assert (out["joints"][f, :n[f]] - ref["joints"][f, :n[f]]).abs().max() < 2e-2
    small = synth.generate(2 * 4, 5, 4, 20260105, chain_len=4)
    hs = HotPath(small["K"], small["Rt"])
    check_chain_flags(run_chains_fused(hs, torch.from_numpy(small["kps25"]).cuda(), torch.from_numpy(small["counts"]).cuda(), 4))


def test_launches_in_flight_on_several_streams_do_not_interact():
    """bench.py keeps two steps in flight on alternating streams (the head of one launch fills the slots its predecessor's slowest
    chains leave idle).  Three launches on three streams, more workgroups than slots in every one of them: each must return exactly
    what it returns alone, and no hand-over may time out (a waiting workgroup's predecessor is dispatched before it in its own launch,
    whatever the other launches do)."""
    from multiview_motion_capture_amd import synth
    from multiview_motion_capture_amd.pipeline import HotPath
    from multiview_motion_capture_amd.tracker import check_chain_flags, run_chains_fused
    B, L = 200, 8
    sets = []
    for s in range(3):
        data = synth.generate(B * L, 5, 4, 20260110 + s, chain_len=L)
        hp = HotPath(data["K"], data["Rt"])
        sets.append((hp, torch.from_numpy(data["kps25"]).cuda(), torch.from_numpy(data["counts"]).cuda()))
    alone = []
    for hp, kps, cnt in sets:
        r = run_chains_fused(hp, kps, cnt, L)
        torch.cuda.synchronize()
        alone.append({k: r[k].clone() for k in ("params", "joints", "meta", "n_tracks")})
    streams = [torch.cuda.Stream() for _ in sets]
    torch.cuda.synchronize()
    together = []
    for rep in range(2):
        for st, (hp, kps, cnt) in zip(streams, sets):
            with torch.cuda.stream(st):
                together.append(run_chains_fused(hp, kps, cnt, L))
    torch.cuda.synchronize()
    for i, r in enumerate(together):
        check_chain_flags(r)
        ref = alone[i % 3]
        for k in ("meta", "n_tracks"):
            assert torch.equal(r[k], ref[k]), (i, k)
        for k in ("params", "joints"):
            assert torch.equal(torch.nan_to_num(r[k]), torch.nan_to_num(ref[k])), (i, k)


def test_throughput_build_and_latency_build_give_the_same_bits():
    """mvmc_chain_run launches the SMALL layout's 256-register LATENCY build (csrc/mvmc_chain_lat.hip) when a call has at most two
    workgroups per CU and the 128-register THROUGHPUT build (csrc/mvmc_chain.hip, four workgroups per CU) above that.  The same source
    at different batch sizes: 1,024 chain-frames in one call (throughput build) against the same chains in two calls of 512 (latency
    build) -- tables, iteration counts and solver records must be equal bit for bit."""
    from multiview_motion_capture_amd import synth
    from multiview_motion_capture_amd.pipeline import HotPath
    from multiview_motion_capture_amd.tracker import check_chain_flags, run_chains_fused
    d = torch.device("cuda:0")
    cus = torch.cuda.get_device_properties(d).multi_processor_count
    L = 16
    n_chains = 2 * ((2 * cus) // L)            # the whole: more than 2 x CUs workgroups; each half: at most 2 x CUs
    assert n_chains * L > 2 * cus >= (n_chains // 2) * L
    data = synth.generate(n_chains * L, 5, 4, 20260103, chain_len=L, occlusion=0.03, spurious=0.2)
    hp = HotPath(data["K"], data["Rt"], device=d)
    kps, cnt = torch.from_numpy(data["kps25"]).to(d), torch.from_numpy(data["counts"]).to(d)
    whole = run_chains_fused(hp, kps, cnt, L, want_info=True)
    h = (n_chains // 2) * L
    halves = [run_chains_fused(hp, kps[s].contiguous(), cnt[s].contiguous(), L, want_info=True) for s in (slice(0, h), slice(h, 2 * h))]
    torch.cuda.synchronize()
    for r in [whole] + halves:
        check_chain_flags(r)
    for k in ("params", "joints", "meta", "n_tracks", "als_iters", "ik_info"):
        a = torch.nan_to_num(whole[k].double())
        b = torch.nan_to_num(torch.cat([halves[0][k], halves[1][k]]).double())
        assert torch.equal(a, b), f"the two builds differ in {k}"
