"""Size-independent properties at BASELINE.json's full size (10 k frames, C5 P4 J25), where the CPU oracle
is too slow to run: determinism, shard invariance (what the multi-GPU split relies on), agreement of the
association with the generator's ground truth, and 3-D accuracy against the ground-truth joints."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
F, C, P, L = 10000, 5, 4, 16


@pytest.fixture(scope="module")
def full():
    from multiview_motion_capture_amd import synth
    from multiview_motion_capture_amd.pipeline import HotPath
    from multiview_motion_capture_amd.tracker import run_chains
    data = synth.generate(F, C, P, 20260103, chain_len=L)
    d = torch.device("cuda:0")
    hp = HotPath(data["K"], data["Rt"], device=d)
    kps = torch.from_numpy(data["kps25"]).to(d)
    cnt = torch.from_numpy(data["counts"]).to(d)
    return dict(data=data, hp=hp, kps=kps, cnt=cnt, run_chains=run_chains)


def test_association_matches_ground_truth_identities(full):
    """Every 2-D pose of a frame must land in the cluster of its true person (the generator shuffles the
    person order independently per view and frame)."""
    hp, data = full["hp"], full["data"]
    assoc = hp.associate(full["kps"], full["cnt"])
    lab = assoc["labels"].cpu().numpy().reshape(F, C, P)
    order = data["gt_order"]  # order[f,c,slot] = true person in that slot
    ok = 0
    for f in range(F):
        person_label = -np.ones(P, dtype=int)
        good = True
        for c in range(C):
            for s in range(P):
                p, l = order[f, c, s], lab[f, c, s]
                if l < 0:
                    good = False
                elif person_label[p] < 0:
                    person_label[p] = l
                elif person_label[p] != l:
                    good = False
        good = good and len(set(person_label.tolist())) == P
        ok += int(good)
    print(f"association: {ok}/{F} frames with every pose in its true person's cluster; "
          f"ALS iterations mean {assoc['iters'].float().mean().item():.0f}")
    assert ok >= 0.97 * F


def test_chain_run_is_deterministic_and_shard_invariant(full):
    run_chains, hp, kps, cnt = full["run_chains"], full["hp"], full["kps"], full["cnt"]
    a = run_chains(hp, kps, cnt, L)
    b = run_chains(hp, kps, cnt, L)
    for k in ("params", "joints", "meta", "n_tracks"):
        assert torch.equal(torch.nan_to_num(a[k].double()), torch.nan_to_num(b[k].double())), f"non-deterministic {k}"
    # two shards processed separately == the whole run (chains never interact): the multi-GPU split is exact
    cut = (F // L // 2) * L
    s0 = run_chains(hp, kps[:cut].contiguous(), cnt[:cut].contiguous(), L)
    s1 = run_chains(hp, kps[cut:].contiguous(), cnt[cut:].contiguous(), L)
    for k in ("params", "joints", "meta", "n_tracks"):
        whole = torch.nan_to_num(a[k].double())
        parts = torch.nan_to_num(torch.cat([s0[k], s1[k]]).double())
        assert torch.equal(whole, parts), f"shard-dependent {k}"
    full["chains"] = a


def test_tracks_and_3d_accuracy_against_ground_truth(full):
    a = full.get("chains") or full["run_chains"](full["hp"], full["kps"], full["cnt"], L)
    n = a["n_tracks"].cpu().numpy()
    meta = a["meta"].cpu().numpy()
    joints = a["joints"].cpu().numpy()
    gt = full["data"]["gt_joints"]  # (F,P,18,3)
    assert (n == P).mean() > 0.97, "almost every frame tracks exactly P people"
    # a chain's tracklets keep their identity: hits grow by one per frame inside a chain
    last = np.arange(F) % L == L - 1
    full_len = (meta[last][:, :P, 2] == L).mean()
    errs = []
    for f in range(0, F, 7):
        if n[f] != P:
            continue
        d = np.linalg.norm(joints[f, :P, None] - gt[f][None], axis=-1).mean(axis=-1)  # (track, person)
        errs.append(d.min(axis=1))
    errs = np.concatenate(errs)
    print(f"tracks: {100 * (n == P).mean():.2f}% frames with {P} tracks; {100 * full_len:.2f}% of tracklets span their "
          f"whole chain; mean joint error vs ground truth: median {np.median(errs) * 100:.2f} cm, p95 "
          f"{np.quantile(errs, 0.95) * 100:.2f} cm")
    assert full_len > 0.9
    assert np.median(errs) < 0.05


def test_stream_groups_do_not_change_results(full):
    """run_chains(n_groups=G) advances G groups of chains on separate HIP streams; chains are independent, so
    the tracklet tables must be identical to the single-stream run."""
    run_chains, hp, kps, cnt = full["run_chains"], full["hp"], full["kps"], full["cnt"]
    n = 64 * L
    a = run_chains(hp, kps[:n].contiguous(), cnt[:n].contiguous(), L)
    b = run_chains(hp, kps[:n].contiguous(), cnt[:n].contiguous(), L, n_groups=3)
    torch.cuda.synchronize()
    for k in ("params", "joints", "meta", "n_tracks", "n_dead"):
        assert torch.equal(torch.nan_to_num(a[k].double()), torch.nan_to_num(b[k].double())), k


def test_persistent_chain_kernel_at_full_size(full):
    """The benchmark's default path (mvmc_chain_run, one launch for the 625 chains) at BASELINE's size: identical to the
    launch-per-stage run, deterministic, and shard invariant (two half-shards == the whole: what the multi-GPU split
    relies on)."""
    from multiview_motion_capture_amd.tracker import run_chains_fused
    hp, kps, cnt = full["hp"], full["kps"], full["cnt"]
    a = full.get("chains") or full["run_chains"](hp, kps, cnt, L)
    f1 = run_chains_fused(hp, kps, cnt, L)
    f2 = run_chains_fused(hp, kps, cnt, L)
    cut = (F // L // 2) * L
    s0 = run_chains_fused(hp, kps[:cut].contiguous(), cnt[:cut].contiguous(), L)
    s1 = run_chains_fused(hp, kps[cut:].contiguous(), cnt[cut:].contiguous(), L)
    torch.cuda.synchronize()
    from multiview_motion_capture_amd.tracker import check_chain_flags
    for r in (f1, f2, s0, s1):
        check_chain_flags(r)                     # no time-out, no graph-size word
        assert int(r["void"].max()) == 0         # ... and no chain with a capacity word (the launch's word is their OR)
    assert not a["overflow"].any()
    for k in ("params", "joints", "meta", "n_tracks"):
        whole = torch.nan_to_num(f1[k].double())
        assert torch.equal(whole, torch.nan_to_num(a[k].double())), f"fused differs from staged: {k}"
        assert torch.equal(whole, torch.nan_to_num(f2[k].double())), f"non-deterministic {k}"
        assert torch.equal(whole, torch.nan_to_num(torch.cat([s0[k], s1[k]]).double())), f"shard-dependent {k}"
    assert torch.equal(f1["n_dead"], a["n_dead"])
