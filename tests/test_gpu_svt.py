"""match_svt on the device (mvmc_svt_associate + mvmc_closure_labels) against the vectors recorded from the reference
(tests/golden/svt_cases.npz: Shelf affinities in float32 and float64, two 64-node C8 P8 graphs) and against the oracle's X."""
import numpy as np
import pytest
import torch

import oracle_np as o
from conftest import load_golden

pytestmark = pytest.mark.gpu


def _cases():
    g = load_golden("svt_cases.npz")
    return g, sorted({k[:-2] for k in g.files if k.endswith("_S")})


def test_match_svt_reference_api_golden():
    from multiview_motion_capture_amd import mv_association as mva
    g, names = _cases()
    for nm in names:
        S, dim = g[nm + "_S"], g[nm + "_dim"]
        S_in = S.copy()
        mm, xb = mva.match_svt(S_in, [int(v) for v in dim])
        assert np.array_equal(S_in, S), "the argument is not modified"
        assert xb.dtype == bool and mm.dtype == bool
        assert np.array_equal(xb, g[nm + "_x_bin"]), nm
        assert np.array_equal(mm, g[nm + "_match_mat"]), nm


def test_svt_batched_iterations_and_X_vs_oracle():
    from multiview_motion_capture_amd import device as dev
    g, names = _cases()
    d = torch.device("cuda:0")
    for dt, tol_x in ((np.float64, 1e-9), (np.float32, 2e-4)):
        sel = [nm for nm in names if nm.endswith("f64" if dt == np.float64 else "f32")]
        N = max(g[nm + "_S"].shape[0] for nm in sel)
        G = max(len(g[nm + "_dim"]) - 1 for nm in sel)
        S = np.zeros((len(sel), N, N), dtype=dt)
        cnt = np.zeros((len(sel), G), dtype=np.int32)
        for k, nm in enumerate(sel):
            n = g[nm + "_S"].shape[0]
            S[k, :n, :n] = g[nm + "_S"]
            c = np.diff(g[nm + "_dim"])
            cnt[k, :len(c)] = c
        res = dev.svt_associate(torch.from_numpy(S).to(d), torch.from_numpy(cnt).to(d), g_max=int(cnt.max()), want_x=True)
        it, X, xb = res["iters"].cpu().numpy(), res["X"].cpu().numpy(), res["x_bin"].cpu().numpy()
        lab = res["labels"].cpu().numpy()
        for k, nm in enumerate(sel):
            n = g[nm + "_S"].shape[0]
            assert it[k] == int(g[nm + "_svd_calls"]), (nm, it[k])
            mm_o, xb_o, info = o.match_svt(g[nm + "_S"], g[nm + "_dim"], return_info=True)
            assert np.array_equal(xb[k, :n, :n].astype(bool), xb_o), nm
            assert (xb[k, n:, :] == 0).all() and (xb[k, :, n:] == 0).all()
            # float64: the oracle's SVD and the Jacobi eigendecomposition agree to rounding; float32 inputs: the oracle iterates in
            # float32 (as torch does), the device in float64
            assert np.abs(X[k, :n, :n] - info["X"]).max() <= tol_x, (nm, np.abs(X[k, :n, :n] - info["X"]).max())
            assert np.array_equal(lab[k, :n], o.cluster_labels(mm_o, n)), nm


def test_svt_options_and_errors():
    from multiview_motion_capture_amd import device as dev, mv_association as mva
    g, names = _cases()
    nm = "shelf295_f64"
    S, dim = g[nm + "_S"], [int(v) for v in g[nm + "_dim"]]
    for kw, okw in ((dict(dual_stochastic_SVT=False), dict(dual_stochastic=False)), (dict(maxIter=3), dict(max_iter=3)),
                    (dict(_lambda=20, mu=32, alpha=0.2), dict(lam=20, mu=32, alpha=0.2)), (dict(tol=5e-2), dict(tol=5e-2))):
        mm, xb = mva.match_svt(S.copy(), dim, **kw)
        mm_o, xb_o = o.match_svt(S, dim, **okw)
        assert np.array_equal(xb, xb_o) and np.array_equal(mm, mm_o.astype(bool)), kw
    with pytest.raises(ValueError):
        mva.match_svt(S.copy(), dim, pselect=0)
    with pytest.raises(ValueError):
        mva.match_svt(S.copy(), dim[:-1])
    d = torch.device("cuda:0")
    with pytest.raises(ValueError):   # more than 64 nodes
        dev.svt_associate(torch.zeros((1, 65, 65), dtype=torch.float64, device=d), torch.full((1, 5), 13, dtype=torch.int32, device=d), 13)
    # an empty graph in the batch runs no iteration and yields no pair
    cnt = torch.tensor([[0, 0, 0, 0, 0], [4, 4, 4, 4, 4]], dtype=torch.int32, device=d)
    Sb = torch.zeros((2, 20, 20), dtype=torch.float64, device=d)
    Sb[1] = torch.from_numpy(S).to(d)
    res = dev.svt_associate(Sb, cnt, 4)
    assert int(res["iters"][0]) == 0 and int(res["x_bin"][0].sum()) == 0 and (res["labels"][0] == -1).all()
