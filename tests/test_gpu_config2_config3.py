"""GPU: BASELINE.json configs 2 and 3 at their full sizes (the other two single-GPU configurations besides the benchmark's config 4).
  config 2: synthetic 10 k frames, C5 P1 J25, triangulation only (seed 20260101) -- ingest + DLT
  config 3: synthetic 10 k frames, C5 P4, affinity + ALS + DLT (seed 20260102)
Each: the oracle on a 64-frame subset (1e-4), and size-independent properties on all 10 k frames."""
import numpy as np
import pytest
import torch

import oracle_np as o
from helpers import frame_nodes, oracle_ingest

pytestmark = pytest.mark.gpu
F = 10000


def test_config2_triangulation_only_10k_c5p1():
    from multiview_motion_capture_amd import device as dev, synth
    C, P = 5, 1
    data = synth.generate(F, C, P, 20260101)
    d = torch.device("cuda:0")
    kps, cnt, Pm = torch.from_numpy(data["kps25"]).to(d), torch.from_numpy(data["counts"]).to(d), torch.from_numpy(data["P"]).to(d)
    k17, c17 = dev.ingest(kps, cnt)
    mem = (torch.arange(F, device=d, dtype=torch.int32)[:, None] * C + torch.arange(C, device=d, dtype=torch.int32)[None]).contiguous()
    pts = dev.dlt(k17, Pm, mem)                 # one cluster per frame: the person's five views
    torch.cuda.synchronize()
    pts = pts.cpu().numpy()
    assert pts.shape == (F, 17, 4) and np.isfinite(pts).all() and (c17.cpu().numpy() == 1).all()
    # oracle on a 64-frame subset: DLT within 1e-4 relative (north star), the mean score exactly
    k17_o, _ = oracle_ingest(data["kps25"][:64].astype(np.float64), data["counts"][:64])
    worst = 0.0
    for f in range(64):
        ref = o.triangulate_groups(data["P"], [k17_o[f, c, 0] for c in range(C)], 0.01, False)
        worst = max(worst, np.abs(pts[f, :, :3] - ref[:, :3]).max() / np.abs(ref[:, :3]).max())
        assert np.array_equal(pts[f, :, 3], ref[:, 3])
    print("config 2: DLT vs oracle on 64 frames, worst relative error %.2e" % worst)
    assert worst < 1e-6       # (north star: 1e-4; inverse iteration on the normal matrix lands at ~1e-10 of LAPACK's SVD)
    # the one-pass form (mvmc_ingest_dlt: the 17-joint tensor stays in LDS) gives the same numbers bit for bit
    fused, c_f = dev.ingest_dlt(kps, cnt, Pm, mem.view(F, 1, C), want_counts=True)
    assert torch.equal(fused.view(F, 17, 4), torch.from_numpy(pts).to(d)) and torch.equal(c_f, c17)
    # float32 points out (mvmc_ingest_dlt_f32: one 16-byte store per point, SURVEY 8(d)'s I/O -- what bench.py times): the float64
    # result rounded ONCE, hence within 1e-6 of the oracle like it
    f32, c_32 = dev.ingest_dlt(kps, cnt, Pm, mem.view(F, 1, C), want_counts=True, out_dtype=torch.float32)
    assert f32.dtype == torch.float32 and torch.equal(f32, fused.float()) and torch.equal(c_32, c17)
    w32 = max(np.abs(f32[f, 0, :, :3].cpu().numpy() - o.triangulate_groups(data["P"], [k17_o[f, c, 0] for c in range(C)], 0.01, False)[:, :3]).max()
              / np.abs(pts[f, :, :3]).max() for f in range(64))
    print("config 2: float32 points vs oracle on 64 frames, worst relative error %.2e" % w32)
    assert w32 < 1e-6
    k64 = kps.double()[:, :, :, [0, 16, 15, 18, 17, 5, 2, 6, 3, 7, 4, 12, 9, 13, 10, 14, 11]].contiguous()   # COCO-17, float64 input
    assert torch.equal(dev.ingest_dlt(k64, None, Pm, mem.view(F, 1, C)), fused)
    # full size: against the generator's ground truth (2 px noise -> about a centimetre), joints seen by >= 2 views
    gt = data["gt_joints"][:, 0]            # (F,18,3)
    coco_from_skel = {0: 15, 3: 16, 4: 17, 5: 9, 6: 12, 7: 10, 8: 13, 9: 11, 10: 14, 11: 1, 12: 4, 13: 2, 14: 5, 15: 3, 16: 6}
    k17h = k17.cpu().numpy()[:, :, 0]       # (F,C,17,3)
    err = []
    for cj, sj in coco_from_skel.items():
        seen = (k17h[:, :, cj, 2] >= 0.01).sum(axis=1) >= 2
        err.append(np.linalg.norm(pts[seen, cj, :3] - gt[seen, sj], axis=-1))
    err = np.concatenate(err)
    print("config 2 at 10 k frames: joint error vs ground truth median %.2f cm, p99 %.2f cm over %d points" %
          (np.median(err) * 100, np.quantile(err, 0.99) * 100, len(err)))
    assert np.median(err) < 0.02 and np.quantile(err, 0.99) < 0.1
    # shard invariance: two halves == the whole
    h = F // 2
    p0 = dev.dlt(k17[:h].contiguous(), Pm, mem[:h].contiguous())
    p1 = dev.dlt(k17[h:].contiguous(), Pm, (mem[h:] - h * C).contiguous())
    assert torch.equal(torch.cat([p0, p1]), torch.from_numpy(pts).to(d))


def test_config3_association_and_triangulation_10k_c5p4():
    from multiview_motion_capture_amd import synth
    from multiview_motion_capture_amd.pipeline import HotPath
    C, P = 5, 4
    data = synth.generate(F, C, P, 20260102)
    d = torch.device("cuda:0")
    hp = HotPath(data["K"], data["Rt"], device=d)
    kps, cnt = torch.from_numpy(data["kps25"]).to(d), torch.from_numpy(data["counts"]).to(d)
    assoc = hp.associate(kps, cnt, want_mats=True)
    tri = hp.triangulate(assoc)
    torch.cuda.synchronize()
    lab = assoc["labels"].cpu().numpy()
    mem, nm, pts = tri["members"].cpu().numpy(), tri["n_members"].cpu().numpy(), tri["pts3d"].cpu().numpy()
    # oracle on a 64-frame subset: association exact, DLT of every cluster within 1e-4
    k17_o, cnt_o = oracle_ingest(data["kps25"][:64].astype(np.float64), data["counts"][:64])
    F_o = hp.F.cpu().numpy()
    worst = 0.0
    for f in range(64):
        pts_o, _, dim, q = frame_nodes(k17_o[f], cnt_o[f])
        D_o, S_o = o.geometry_affinity(pts_o, F_o, dim)
        assert np.array_equal(assoc["D"][f].cpu().numpy()[:len(q), :len(q)], D_o)
        mm_o, xb_o = o.match_als(assoc["S"][f, :len(q), :len(q)].cpu().numpy(), dim)
        lab_o = o.cluster_labels(mm_o, len(q))
        assert np.array_equal(assoc["x_bin"][f, :len(q), :len(q)].cpu().numpy().astype(bool), xb_o)
        assert np.array_equal(lab[f, :len(q)], lab_o)
        for k in range(lab_o.max() + 1):
            nodes = np.nonzero(lab_o == k)[0]
            if len(nodes) < 2:
                continue
            cams = [q[i] // P for i in nodes]
            ref = o.triangulate_groups(data["P"][cams], [k17_o[f, q[i] // P, q[i] % P] for i in nodes], 0.01, False)
            got = pts[f, k]
            assert [int(m % (C * P)) for m in mem[f, k, :len(nodes)]] == [q[i] for i in nodes]
            worst = max(worst, np.abs(got[:, :3] - ref[:, :3]).max() / np.abs(ref[:, :3]).max())
    print("config 3: association exact on 64 frames; DLT of the clusters vs oracle worst relative error %.2e" % worst)
    assert worst < 1e-4
    # full size: identities against the generator's truth and 3-D accuracy of the triangulated clusters
    order, gt = data["gt_order"], data["gt_joints"]
    good = 0
    err = []
    for f in range(0, F, 5):
        l = lab[f].reshape(C, P)
        pl = {}
        ok = True
        for c in range(C):
            for s in range(P):
                if l[c, s] < 0 or pl.setdefault(order[f, c, s], l[c, s]) != l[c, s]:
                    ok = False
        ok = ok and len(set(pl.values())) == P
        good += ok
        if ok:
            for p, k in pl.items():
                if nm[f, k] >= 2:
                    err.append(np.linalg.norm(pts[f, k, 11, :3] - gt[f, p, 1]))      # left hip (COCO 11 = skeleton joint 1)
    err = np.array(err)
    print("config 3 at 10 k frames: %d / %d sampled frames with every pose in its person's cluster; hip error median %.2f cm" %
          (good, len(range(0, F, 5)), np.median(err) * 100))
    assert good >= 0.97 * len(range(0, F, 5)) and np.median(err) < 0.02


@pytest.mark.parametrize("C,P,K,V,J,F", [(5, 1, 1, 5, 25, 1003), (8, 2, 3, 8, 25, 517), (3, 4, 5, 3, 17, 260), (6, 3, 11, 4, 25, 97),
                                         (4, 2, 12, 4, 25, 64)])
def test_one_pass_triangulation_equals_the_two_kernels_on_ragged_shapes(C, P, K, V, J, F):
    """mvmc_ingest_dlt's float32 pipeline (loader wave + triangulating waves, LDS-DMA gathers, groups that do not divide the frame
    count) against mvmc_ingest + mvmc_dlt, bit for bit, where the shapes are NOT the benchmark's: several people per view with ragged
    counts and poses the filter drops, several clusters per frame with holes (-1) and fewer or more than five views, COCO-17 input,
    the unrolled (V <= 5) and the generic view loop, more clusters than the pipeline holds (K = 12: the first one-pass kernel)."""
    from multiview_motion_capture_amd import device as dev
    rng = np.random.default_rng(1000 * C + 10 * P + K)
    d = torch.device("cuda:0")
    kps = rng.uniform(50, 900, (F, C, P, J, 3)).astype(np.float32)
    kps[..., 2] = rng.uniform(0, 1, (F, C, P, J)).astype(np.float32)
    kps[rng.uniform(size=(F, C, P, J)) < 0.1] = 0.0                               # dropped joints, OpenPose style
    kps[rng.uniform(size=(F, C, P)) < 0.1] *= np.float32(0.001)                   # a pose the filter rejects (tiny box)
    cnt = rng.integers(0, P + 1, (F, C)).astype(np.int32)
    Pm = rng.normal(size=(C, 3, 4)) * np.array([1000.0, 1000.0, 1.0])[None, :, None]
    Pm[:, 2, 3] += 4.0
    kd, cd, Pd = torch.from_numpy(kps).to(d), torch.from_numpy(cnt).to(d), torch.from_numpy(Pm).to(d)
    k17, c17 = dev.ingest(kd, cd)
    c17h = c17.cpu().numpy()
    # clusters: random subsets of the frame's kept poses (ingest numbering (f C + c) P + slot), padded with -1, some empty
    mem = -np.ones((F, K, V), dtype=np.int32)
    for f in range(F):
        pool = [(f * C + c) * P + s for c in range(C) for s in range(c17h[f, c])]
        for k in range(K):
            n = int(rng.integers(0, V + 1))
            if pool and n:
                pick = rng.choice(len(pool), size=min(n, len(pool)), replace=False)
                where = np.sort(rng.choice(V, size=len(pick), replace=False))
                mem[f, k, where] = np.array(pool)[pick]
    md = torch.from_numpy(mem).to(d)
    two = dev.dlt(k17, Pd, md.view(F * K, V)).view(F, K, 17, 4)
    one, c_one = dev.ingest_dlt(kd, cd, Pd, md, want_counts=True)
    torch.cuda.synchronize()
    assert torch.equal(c_one, c17)
    a, b = one.cpu().numpy(), two.cpu().numpy()
    assert np.array_equal(a, b, equal_nan=True), (np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))[:5])
    assert np.isfinite(a[..., 3]).sum() > 0
    # the float32-output entry: the same numbers rounded once where the pipeline holds the shape, a clean refusal where it does not
    if K * 17 <= 192:
        o32 = dev.ingest_dlt(kd, cd, Pd, md, out_dtype=torch.float32)
        assert np.array_equal(o32.cpu().numpy(), a.astype(np.float32), equal_nan=True)
    else:
        from multiview_motion_capture_amd._cabi import MvmcError
        with pytest.raises(MvmcError):
            dev.ingest_dlt(kd, cd, Pd, md, out_dtype=torch.float32)
