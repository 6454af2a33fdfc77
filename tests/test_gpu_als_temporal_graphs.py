"""ALS on the spatio-temporal graphs of the benchmark workload (24 nodes: 4 tracklets + 5 views x 4 people, rank 8 -- the case the
solver-wave kernel serves) against the oracle's match_als on the same float64 affinities: labels bit-exact, iteration counts equal up to
the rounding of the stopping thresholds (mv_association.py:222-318)."""
import numpy as np
import pytest
import torch

import oracle_np as o

pytestmark = pytest.mark.gpu


def test_temporal_graphs_labels_and_iterations():
    from multiview_motion_capture_amd import device as dev, synth
    from multiview_motion_capture_amd.pipeline import HotPath
    from multiview_motion_capture_amd.tracker import ChainTracker
    L, B = 5, 96
    data = synth.generate(B * L, 5, 4, 20260103, chain_len=L)
    d = torch.device("cuda:0")
    hp = HotPath(data["K"], data["Rt"], device=d)
    kps17, cnt = dev.ingest(torch.from_numpy(data["kps25"]).to(d), torch.from_numpy(data["counts"]).to(d))
    k4, c4 = kps17.view(B, L, 5, 4, 17, 3), cnt.view(B, L, 5)
    tr = ChainTracker(hp, B, 4)
    n_graphs, n_diff, worst = 0, 0, 0
    for t in range(L):
        out = tr.step(k4[:, t].contiguous(), c4[:, t].contiguous(), want_debug=True)
        if t == 0:
            continue
        W, gc = out["W"].cpu().numpy(), out["group_counts"].cpu().numpy()
        lab, it = out["st"]["labels"].cpu().numpy(), out["st"]["iters"].cpu().numpy()
        xb = out["st"]["x_bin"].cpu().numpy()
        for b in range(B):
            dim = [0] + np.cumsum(gc[b]).tolist()
            n = dim[-1]
            assert n == 24
            mm_o, xb_o, it_o = o.match_als(W[b, :n, :n], dim, return_iters=True)
            assert np.array_equal(xb[b, :n, :n].astype(bool), xb_o), (t, b)
            assert np.array_equal(lab[b, :n], o.cluster_labels(mm_o, n)), (t, b)
            n_graphs += 1
            n_diff += int(it[b] != it_o)
            worst = max(worst, abs(int(it[b]) - it_o))
    # the counts are decided by thresholds on fp64 norms (stop, and the doubling / halving of mu); rounding moves a few of them
    print(f"{n_graphs} temporal graphs: {n_diff} iteration counts differ from the oracle's, worst by {worst}")
    assert n_diff <= n_graphs // 10 and worst <= 2, (n_diff, worst)


@pytest.mark.parametrize("sizes,dtype", [
    ([4, 4, 4, 4, 4, 4, 4], np.float64),      # n = 28, rank 8: the four-wave form with partial sums (n > 24)
    ([4, 4, 4, 4, 4, 4, 4], np.float32),
    ([8, 4, 4, 4, 4, 4], np.float64),         # n = 28, rank 16
    ([3, 6, 5, 6, 6, 6], np.float64),         # n = 32, rank 12 (padded to 16), ragged groups
    ([4, 5, 3, 4, 4], np.float32),            # n = 20, rank 10 > 8: not the solver-wave form
    ([2, 3, 0, 4, 3], np.float64),            # n = 12 with an empty group: the solver-wave form on a small graph
])
def test_workgroup_als_variants_vs_oracle(sizes, dtype):
    """Every variant behind mvmc_als_associate's workgroup path on block-structured affinities: labels exact, iterations +-2."""
    from multiview_motion_capture_amd import device as dev
    rng = np.random.default_rng(20260107)
    n, B = int(np.sum(sizes)), 12
    dim = [0] + np.cumsum(sizes).tolist()
    W = np.zeros((B, n, n), dtype=dtype)
    for b in range(B):
        ident = np.concatenate([rng.permutation(max(sizes))[:s] for s in sizes])
        same = ident[:, None] == ident[None, :]
        A = np.where(same, rng.uniform(0.55, 1.0, (n, n)), rng.uniform(0.0, 0.4, (n, n)))
        A = 0.5 * (A + A.T)
        for g in range(len(sizes)):
            A[dim[g]:dim[g + 1], dim[g]:dim[g + 1]] = 0.0
        W[b] = A.astype(dtype)
    d = torch.device("cuda:0")
    cnt = torch.tensor([sizes] * B, dtype=torch.int32, device=d)
    res = dev.als_associate(torch.from_numpy(W).to(d), cnt, g_max=max(sizes), want_mats=True)
    lab, it, xb = res["labels"].cpu().numpy(), res["iters"].cpu().numpy(), res["x_bin"].cpu().numpy()
    n_diff = 0
    for b in range(B):
        mm_o, xb_o, it_o = o.match_als(W[b], dim, return_iters=True)
        assert np.array_equal(xb[b].astype(bool), xb_o), b
        assert np.array_equal(lab[b], o.cluster_labels(mm_o, n)), b
        assert abs(int(it[b]) - it_o) <= 2, (b, it[b], it_o)
        n_diff += int(it[b] != it_o)
    assert n_diff <= 2, n_diff
