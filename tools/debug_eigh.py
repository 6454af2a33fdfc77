import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import oracle_np as o, trf_np as t
from test_gpu_eigh import _run
g = np.load(os.path.join(ROOT, "tests/golden/ik_cases.npz"))
i = int(np.nonzero(~g["cold"])[0][0]); v = int(g["n_views"][i])
obs = np.array([o.add_mid_spine(p) for p in g["poses"][i, :v]])[:, o.IK_OBS_IDX, :]; projs = np.asarray(g["projs"][i, :v])
x = g["s2_x0"][i]
J = t.ik_jacobian(x[:3], x[3:57], x[57:], obs, projs, True); f = o.ik_residual(x[:3], x[3:57], x[57:], obs, projs)
act = np.nonzero(np.abs(J).max(axis=0) > 0)[0]
Ja = np.zeros((J.shape[0], 50)); Ja[:, :len(act)] = J[:, act]
A = Ja.T @ Ja; gr = Ja.T @ f
lam, Vt, k0 = _run(A[None], gr[None]); lam, Vt, k0 = lam[0], Vt[0], int(k0[0])
w, V = np.linalg.eigh(A)
np.set_printoptions(precision=3, linewidth=200)
print("n active", len(act), "k0", k0, "lmax %.3e" % w[-1])
print("numpy smallest eigs", w[:8]); print("gpu lam first", lam[:8])
print("numpy |V^T g| first 8", np.abs(V.T @ gr)[:8]); print("gpu |Vt g| first 8", np.abs(Vt @ gr)[:8])
print("|g| %.3e  |P0 g| numpy %.3e  gpu u.g %.3e" % (np.linalg.norm(gr), np.linalg.norm(V[:, :k0] @ (V[:, :k0].T @ gr)), Vt[0] @ gr))
Vr = Vt[k0:]
print("resolved orth err", np.abs(Vr @ Vr.T - np.eye(len(Vr))).max(), "| g - sum resolved| ", np.linalg.norm(gr - Vr.T @ (Vr @ gr)))
print("numpy complement", np.linalg.norm(gr - V[:, k0:] @ (V[:, k0:].T @ gr)))
