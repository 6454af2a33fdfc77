"""Config 5 geometry (C = 8 views, P = 8 people, n = 64 graph nodes, rank 16) at a small frame count:
the generic kernel variants against the oracle and the generator's ground truth."""
import os

import numpy as np
import pytest
import torch

import oracle_np as o
from helpers import frame_nodes, oracle_ingest

pytestmark = pytest.mark.gpu
F, C, P, L = 64, 8, 8, 16


@pytest.fixture(scope="module")
def c5():
    from multiview_motion_capture_amd import synth
    from multiview_motion_capture_amd.pipeline import HotPath
    data = synth.generate(F, C, P, 20260104, chain_len=L, dtype=np.float64)
    d = torch.device("cuda:0")
    hp = HotPath(data["K"], data["Rt"], device=d)
    return dict(data=data, hp=hp, kps=torch.from_numpy(data["kps25"]).to(d), cnt=torch.from_numpy(data["counts"]).to(d))


def test_spatial_association_n64_vs_oracle_and_truth(c5):
    hp, data = c5["hp"], c5["data"]
    assoc = hp.associate(c5["kps"], c5["cnt"], want_mats=True)
    k17_o, cnt_o = oracle_ingest(data["kps25"][:2], data["counts"][:2])
    assert np.array_equal(assoc["kps17"][:2].cpu().numpy(), k17_o)
    F_o = o.pairwise_f_mats(data["K"], data["Rt"])
    for f in range(2):
        pts, _, dim, _ = frame_nodes(k17_o[f], cnt_o[f])
        D_o, S_o = o.geometry_affinity(pts, hp.F.cpu().numpy(), dim)
        n = len(pts)
        assert n == 64
        assert np.array_equal(assoc["D"][f, :n, :n].cpu().numpy(), D_o)
        mm_o, xb_o, it_o = o.match_als(assoc["S"][f, :n, :n].cpu().numpy(), dim, return_iters=True)
        assert np.array_equal(assoc["x_bin"][f, :n, :n].cpu().numpy().astype(bool), xb_o)
        assert np.array_equal(assoc["labels"][f, :n].cpu().numpy(), o.cluster_labels(mm_o, n))
        assert int(assoc["iters"][f]) == it_o
    lab = assoc["labels"].cpu().numpy().reshape(F, C, P)
    order = data["gt_order"]
    ok = 0
    for f in range(F):
        pl = -np.ones(P, dtype=int)
        good = True
        for c in range(C):
            for s in range(P):
                p, l = order[f, c, s], lab[f, c, s]
                if l < 0 or (pl[p] >= 0 and pl[p] != l):
                    good = False
                pl[p] = l
        ok += int(good and len(set(pl.tolist())) == P)
    print(f"C8 P8 association: {ok}/{F} frames perfect")
    assert ok >= 0.9 * F


def test_chains_c8p8(c5):
    from multiview_motion_capture_amd.tracker import run_chains
    a = run_chains(c5["hp"], c5["kps"], c5["cnt"], L, t_max=12)
    b = run_chains(c5["hp"], c5["kps"], c5["cnt"], L, t_max=12)
    assert torch.equal(torch.nan_to_num(a["joints"]), torch.nan_to_num(b["joints"]))
    n = a["n_tracks"].cpu().numpy()
    joints, gt = a["joints"].cpu().numpy(), c5["data"]["gt_joints"]
    errs = []
    for f in range(F):
        if n[f] == P:
            d = np.linalg.norm(joints[f, :P, None] - gt[f][None], axis=-1).mean(axis=-1)
            errs.append(d.min(axis=1))
    errs = np.concatenate(errs)
    print(f"C8 P8 chains: {100 * (n == P).mean():.1f}% frames with {P} tracks; median joint error {np.median(errs) * 100:.2f} cm")
    assert (n == P).mean() > 0.9 and np.median(errs) < 0.05


def test_temporal_graph_n72_vs_oracle(c5):
    """match_spatial_time at config-5 size (8 tracklets + 64 poses = 72 nodes, generic ALS variant): the device's graph
    for the second frame of a chain -- distances, affinity, X_bin, cluster labels -- against the oracle, both fed the
    tracklets the device produced on the first frame."""
    import tracker_np as tk
    from multiview_motion_capture_amd import device as dev
    from multiview_motion_capture_amd.tracker import ChainTracker
    hp, data = c5["hp"], c5["data"]
    d = c5["kps"].device
    T = 12
    kps17, cnt = dev.ingest(c5["kps"][:2].contiguous(), c5["cnt"][:2].contiguous())
    tr = ChainTracker(hp, 1, P, t_max=T)
    tr.step(kps17[0:1].contiguous(), cnt[0:1].contiguous())
    nt = int(tr.n_tracks[0])
    assert nt == P
    joints0 = tr.joints[0, :nt].cpu().numpy()
    out = tr.step(kps17[1:2].contiguous(), cnt[1:2].contiguous(), want_debug=True)
    k17, c1 = kps17[1].cpu().numpy(), cnt[1].cpu().numpy()
    views = [[k17[c, p] for p in range(c1[c])] for c in range(C)]
    Pm = hp.P.cpu().numpy()
    D_o, dim = o.spatial_time_distance([joints0[k] for k in range(nt)], views, Pm)
    _, S_o = o.spatial_time_affinity(D_o)
    mm_o, xb_o = o.match_als(S_o, dim)
    n = dim[-1]
    assert n == 72 and out["group_counts"][0].cpu().tolist() == np.diff(dim).tolist()
    Dg, Wg = out["D"][0, :n, :n].cpu().numpy(), out["W"][0, :n, :n].cpu().numpy()
    assert np.array_equal(np.isnan(Dg), np.isnan(D_o))
    assert np.nanmax(np.abs(Dg - D_o)) <= 1e-9 * np.nanmax(np.abs(D_o))
    assert np.abs(Wg - S_o).max() < 1e-10
    assert np.array_equal(out["st"]["x_bin"][0, :n, :n].cpu().numpy().astype(bool), xb_o)
    assert np.array_equal(out["st"]["labels"][0, :n].cpu().numpy(), o.cluster_labels(mm_o, n))
    # every tracklet is matched in >= 2 views and continues (status 2), no new tracklets
    assert out["status"][0, :nt].cpu().tolist() == [2] * nt and int(out["n_new"][0]) == 0


def _oracle_als(job):
    W, dim = job
    mm, xb, it = o.match_als(W, dim, return_iters=True)
    return xb, o.cluster_labels(mm, dim[-1]), it


def test_als5_vs_oracle_on_many_graphs():
    """als5_graph (the association of the BIG layout: 61 % of a config-5 chain's cycles) against oracle_np.match_als on the workload's
    own graphs: 96 float32 spatial graphs of 64 nodes (match_spatial at every frame) and 90 float64 temporal graphs of 72 nodes
    (match_spatial_time along six chains) of synthetic C8 P8, seed 20260104: X_bin and labels exact, iteration counts +-2
    (mv_association.py:222-318).  The oracle's 186 graphs run in a process pool on the host."""
    import multiprocessing as mp
    from multiview_motion_capture_amd import device as dev, synth
    from multiview_motion_capture_amd.pipeline import HotPath
    from multiview_motion_capture_amd.tracker import ChainTracker
    B, Fn = 6, 96
    data = synth.generate(Fn, C, P, 20260104, chain_len=L)
    d = torch.device("cuda:0")
    hp = HotPath(data["K"], data["Rt"], device=d)
    kps, cnt = torch.from_numpy(data["kps25"]).to(d), torch.from_numpy(data["counts"]).to(d)
    jobs, got = [], []
    # spatial: every frame on its own (mvmc_als_associate dispatches n <= 72, rank <= 16, few graphs to als5_kernel)
    assoc = hp.associate(kps, cnt, want_mats=True)
    S, xb, lab, it = (assoc[k].cpu().numpy() for k in ("S", "x_bin", "labels", "iters"))
    c_np = assoc["counts"].cpu().numpy() if "counts" in assoc else data["counts"]
    for f in range(Fn):
        dim = [0] + np.cumsum(c_np[f]).tolist()
        n = dim[-1]
        assert n == 64 and S.dtype == np.float32
        jobs.append((S[f, :n, :n], dim))
        got.append((xb[f, :n, :n].astype(bool), lab[f, :n], int(it[f])))
    # temporal: six chains advanced by the per-stage tracker; the graph of every later frame
    kps17, c17 = dev.ingest(kps, cnt)
    k4, c4 = kps17.view(B, L, C, P, 17, 3), c17.view(B, L, C)
    tr = ChainTracker(hp, B, P, t_max=8)
    for t in range(L):
        out = tr.step(k4[:, t].contiguous(), c4[:, t].contiguous(), want_debug=True)
        if t == 0:
            continue
        W, gc = out["W"].cpu().numpy(), out["group_counts"].cpu().numpy()
        lab_t, it_t, xb_t = out["st"]["labels"].cpu().numpy(), out["st"]["iters"].cpu().numpy(), out["st"]["x_bin"].cpu().numpy()
        for b in range(B):
            dim = [0] + np.cumsum(gc[b]).tolist()
            n = dim[-1]
            assert 64 < n <= 72 and W.dtype == np.float64
            jobs.append((W[b, :n, :n].copy(), dim))
            got.append((xb_t[b, :n, :n].astype(bool), lab_t[b, :n], int(it_t[b])))
    tr.check()
    with mp.get_context("fork").Pool(min(16, os.cpu_count() or 1)) as pool:
        exp = pool.map(_oracle_als, jobs, chunksize=2)
    n_diff, worst = 0, 0
    for i, ((xb_g, lab_g, it_g), (xb_o, lab_o, it_o)) in enumerate(zip(got, exp)):
        assert np.array_equal(xb_g, xb_o), i
        assert np.array_equal(lab_g, lab_o), i
        n_diff += int(it_g != it_o)
        worst = max(worst, abs(it_g - it_o))
    n72 = sum(1 for _, dim in jobs if dim[-1] == 72)
    print(f"als5 vs oracle: {len(jobs)} graphs ({Fn} spatial f32 n=64, {len(jobs) - Fn} temporal f64 of which {n72} with n=72): "
          f"{n_diff} iteration counts differ, worst by {worst}")
    assert len(jobs) >= 2 * 64 and worst <= 2 and n_diff <= len(jobs) // 10


def test_chain_kernel_big_layout_is_bit_identical_to_the_staged_path(c5):
    """Config 5 on the persistent chain kernel (mvmc_chain_run, BIG layout: N = 64, N + T = 72, rank 16, 8 views per person): the
    same device functions as the launch-per-stage path, so the same results bit for bit -- and that path is the one the tests above
    and below compare with the oracle."""
    from multiview_motion_capture_amd.tracker import check_chain_flags, run_chains, run_chains_fused
    a = run_chains(c5["hp"], c5["kps"], c5["cnt"], L, t_max=8, want_info=True)
    for parts in (None, 1, 4):
        b = run_chains_fused(c5["hp"], c5["kps"], c5["cnt"], L, t_max=8, want_info=True, parts=parts)
        torch.cuda.synchronize()
        check_chain_flags(b)
        assert torch.equal(a["n_tracks"], b["n_tracks"]) and torch.equal(a["meta"], b["meta"]) and torch.equal(a["n_dead"], b["n_dead"])
        n = a["n_tracks"].cpu().numpy()
        ja, jb, pa, pb = a["joints"].cpu().numpy(), b["joints"].cpu().numpy(), a["params"].cpu().numpy(), b["params"].cpu().numpy()
        for f in range(F):
            assert np.array_equal(ja[f, :n[f]], jb[f, :n[f]]) and np.array_equal(pa[f, :n[f]], pb[f, :n[f]]), f
    joints, gt = b["joints"].cpu().numpy(), c5["data"]["gt_joints"]
    errs = np.concatenate([np.linalg.norm(joints[f, :P, None] - gt[f][None], axis=-1).mean(axis=-1).min(axis=1) for f in range(F) if n[f] == P])
    it = b["als_iters"].cpu().numpy()
    print(f"C8 P8 fused: {100 * (n == P).mean():.1f}% frames with {P} tracks; median joint error {np.median(errs) * 100:.2f} cm; "
          f"ALS iterations heads mean {it[:, 0].mean():.0f}, later frames mean {it[:, 1:].mean():.0f} max {it.max()}")
    assert (n == P).mean() > 0.9 and np.median(errs) < 0.05


def test_temporal_graph_n72_t8_vs_oracle(c5):
    """The 72-node graph of the BIG layout (8 tracklets + 64 poses; the 256-thread generic ALS, n <= 72) against the oracle."""
    from multiview_motion_capture_amd import device as dev
    from multiview_motion_capture_amd.tracker import ChainTracker
    hp = c5["hp"]
    kps17, cnt = dev.ingest(c5["kps"][16:18].contiguous(), c5["cnt"][16:18].contiguous())
    tr = ChainTracker(hp, 1, P, t_max=8)
    tr.step(kps17[0:1].contiguous(), cnt[0:1].contiguous())
    nt = int(tr.n_tracks[0])
    assert nt == P
    joints0 = tr.joints[0, :nt].cpu().numpy()
    out = tr.step(kps17[1:2].contiguous(), cnt[1:2].contiguous(), want_debug=True)
    k17, c1 = kps17[1].cpu().numpy(), cnt[1].cpu().numpy()
    views = [[k17[c, p] for p in range(c1[c])] for c in range(C)]
    D_o, dim = o.spatial_time_distance([joints0[k] for k in range(nt)], views, hp.P.cpu().numpy())
    _, S_o = o.spatial_time_affinity(D_o)
    mm_o, xb_o, it_o = o.match_als(S_o, dim, return_iters=True)
    n = dim[-1]
    assert n == 72
    assert np.array_equal(out["st"]["x_bin"][0, :n, :n].cpu().numpy().astype(bool), xb_o)
    assert np.array_equal(out["st"]["labels"][0, :n].cpu().numpy(), o.cluster_labels(mm_o, n))
    assert abs(int(out["st"]["iters"][0]) - it_o) <= 2


def test_config5_at_full_size_on_the_persistent_kernel():
    """BASELINE config 5 at one GPU's share of it (200 k frames / 8 GPUs = 25,008 frames of C8 P8 in chains of 16) through
    mvmc_chain_run's BIG layout: no hand-over / graph-size / table flag, deterministic, shard invariant (two half-shards == the whole: what
    the multi-GPU split relies on), eight tracklets that keep their identity through their chain, 3-D accuracy against the
    generator's ground truth.  (Parity with the oracle is the small-size tests' above; the CPU oracle needs minutes per chain here.)"""
    from multiview_motion_capture_amd import synth
    from multiview_motion_capture_amd.pipeline import HotPath
    from multiview_motion_capture_amd.tracker import run_chains, run_chains_fused
    from multiview_motion_capture_amd.tracker import check_chain_flags
    Ff = 25008
    data = synth.generate(Ff, C, P, 20260104, chain_len=L)        # BASELINE.json config 5's seed
    d = torch.device("cuda:0")
    hp = HotPath(data["K"], data["Rt"], device=d)
    kps, cnt = torch.from_numpy(data["kps25"]).to(d), torch.from_numpy(data["counts"]).to(d)
    a = run_chains_fused(hp, kps, cnt, L)
    b = run_chains_fused(hp, kps, cnt, L)
    cut = (Ff // L // 2) * L
    s0 = run_chains_fused(hp, kps[:cut].contiguous(), cnt[:cut].contiguous(), L)
    s1 = run_chains_fused(hp, kps[cut:].contiguous(), cnt[cut:].contiguous(), L)
    torch.cuda.synchronize()
    # No word of any kind: with 64 detections per frame match_als now and then merges two people into one cluster of more than eight
    # poses (1 chain in 256) -- round 2 held eight views per person and raised the capacity word there; the view pool of the IK phase
    # holds a cluster of any size now, as the reference does (motion_capture.py:618-626).
    for r in (a, b, s0, s1):
        check_chain_flags(r)
        assert int(r["void"].max()) == 0
    sub = run_chains(hp, kps[:4096].contiguous(), cnt[:4096].contiguous(), L, t_max=a["params"].shape[1])   # (sixteen slots on the BIG layout)
    assert not sub["overflow"].any()
    for k in ("meta", "n_tracks"):
        assert torch.equal(sub[k], a[k][:4096]), f"fused and per-stage paths differ in {k}"
    for k in ("params", "joints", "meta", "n_tracks"):
        whole = torch.nan_to_num(a[k].double())
        assert torch.equal(whole, torch.nan_to_num(b[k].double())), f"non-deterministic {k}"
        assert torch.equal(whole, torch.nan_to_num(torch.cat([s0[k], s1[k]]).double())), f"shard-dependent {k}"
    n = a["n_tracks"].cpu().numpy()
    meta = a["meta"].cpu().numpy()
    joints = a["joints"].cpu().numpy()
    gt = data["gt_joints"]
    last = np.arange(Ff) % L == L - 1
    full_len = (meta[last][:, :P, 2] == L).mean()
    errs = []
    for f in range(0, Ff, 23):
        if n[f] != P:
            continue
        dist = np.linalg.norm(joints[f, :P, None] - gt[f][None], axis=-1).mean(axis=-1)
        errs.append(dist.min(axis=1))
    errs = np.concatenate(errs)
    print(f"config 5, {Ff} frames: {100 * (n == P).mean():.2f}% frames with {P} tracks; {100 * full_len:.2f}% of tracklets span their "
          f"whole chain; mean joint error vs ground truth: median {np.median(errs) * 100:.2f} cm, p95 {np.quantile(errs, 0.95) * 100:.2f} cm")
    assert (n == P).mean() > 0.95 and full_len > 0.9 and np.median(errs) < 0.05
