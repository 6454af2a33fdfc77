"""Golden vectors: every iterate of the reference's trust-region solves.

TEST INFRASTRUCTURE (build container only).  Runs the reference's own ``solve_pose_reproj`` and
``solve_pose_bone_lens_reproj`` (/root/reference/src/inverse_kinematics.py:202-277) through ``oracle/ref_shim.py`` with
SciPy's ``trf_no_bounds`` instrumented from the outside: the names that function looks up in its own module
(``svd``, ``solve_lsq_trust_region``, ``evaluate_quadratic``, ``update_tr_radius``, ``check_termination``;
scipy/optimize/_lsq/trf.py:401-560) are wrapped so that each call's arguments and results are kept -- the arithmetic is
untouched (the script asserts that every traced solve returns bit for bit what the un-instrumented fixtures hold).
Only data is written (inputs and recorded iterates as .npz); no reference source travels.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_trf_traces.py

BLAS threading is left at the container's default (8 threads) ON PURPOSE: ik_cases.npz was recorded that way, and the script asserts
that every traced Shelf solve ends bit for bit where that fixture says.  With OPENBLAS_NUM_THREADS=1 LAPACK's gesdd rounds differently and
the reference's cold solves end up to 3 (!) away in parameter space after 43 evaluations, the warm ones 1e-2 -- the sensitivity that
DESIGN.md's "IK parity" section is about; per STEP (what this fixture is for) the difference is rounding-sized.

Writes tests/golden/ik_trf_traces.npz:
  cases (n = 96): the 64 PoseSolver.solve cases of ik_cases.npz (Shelf: 19 cold, 45 warm; budget 50 / 5 evaluations per
        stage as inverse_kinematics.py:397,400) and 32 clusters of synthetic config 4 from ik_converged.npz (8 cold, 24 warm)
        case_poses (n,6,17,3)  case_projs (n,6,3,4)  case_nviews  case_cold  case_init (n,68)  case_source (0 Shelf, 1 synthetic)
        case_x (n,2,68)  case_cost (n,2)  case_nfev (n,2)  case_njev (n,2)  case_status (n,2)     results of the two stages
  trials (one row per call of solve_lsq_trust_region = one trial step), in solve order:
        t_case, t_stage (0, 1), t_model (index of the Jacobian the trial was made from, within its solve), t_nfev (after the trial)
        t_x (68; stage 0 uses the first 57)   the point the model was built at
        t_cost, t_g (68) = J^T f, t_s (68) singular values of J (descending; trf.py:466)
        t_Delta, t_alpha_in   -> solve_lsq_trust_region (common.py:57-168) ->   t_step (68), t_alpha, t_niter
        t_pred (trf.py:497), t_cost_new, t_actual, t_ratio, t_Delta_new (common.py:222-245), t_status (-1 = none), t_accepted
  jac_trial, jac (k,160|..): the finite-difference Jacobian itself for the first model of every case (pins the restated 2-point rule)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_shim  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
V_PAD = 6
N_SYNTH_COLD, N_SYNTH_WARM = 8, 24


class TrfRecorder:
    """Keeps what trf_no_bounds hands to and gets from its helpers, one record per trial step."""

    NAMES = ("svd", "solve_lsq_trust_region", "evaluate_quadratic", "update_tr_radius", "check_termination")

    def __init__(self):
        import scipy.optimize._lsq.trf as trf_mod
        self.mod = trf_mod
        self.orig = {k: getattr(trf_mod, k) for k in self.NAMES}
        self.solves = []

    def __enter__(self):
        o = self.orig
        rec = self

        def svd(J, **kw):
            r = o["svd"](J, **kw)
            rec.cur["models"].append(dict(J=np.array(J), s=np.array(r[1])))
            return r

        def solve(n, m, uf, s, V, Delta, initial_alpha=None, **kw):
            p, alpha, n_iter = o["solve_lsq_trust_region"](n, m, uf, s, V, Delta, initial_alpha=initial_alpha, **kw)
            rec.cur["trials"].append(dict(model=len(rec.cur["models"]) - 1, Delta=float(Delta), alpha_in=float(initial_alpha),
                                          step=np.array(p), alpha=float(alpha), niter=int(n_iter)))
            return p, alpha, n_iter

        def evalq(J, g, s, **kw):
            v = o["evaluate_quadratic"](J, g, s, **kw)
            t = rec.cur["trials"][-1]
            t["g"] = np.array(g)
            t["pred"] = float(-v)
            return v

        def update(Delta, actual, pred, step_norm, bound_hit):
            Delta_new, ratio = o["update_tr_radius"](Delta, actual, pred, step_norm, bound_hit)
            t = rec.cur["trials"][-1]
            t.update(actual=float(actual), ratio=float(ratio), Delta_new=float(Delta_new), step_norm=float(step_norm))
            return Delta_new, ratio

        def check(dF, F, dx_norm, x_norm, ratio, ftol, xtol):
            st = o["check_termination"](dF, F, dx_norm, x_norm, ratio, ftol, xtol)
            t = rec.cur["trials"][-1]
            t["cost"] = float(F)
            t["status"] = -1 if st is None else int(st)
            return st

        for k, f in dict(svd=svd, solve_lsq_trust_region=solve, evaluate_quadratic=evalq, update_tr_radius=update,
                         check_termination=check).items():
            setattr(self.mod, k, f)
        return self

    def __exit__(self, *a):
        for k, f in self.orig.items():
            setattr(self.mod, k, f)

    def begin(self, x0):
        self.cur = dict(x0=np.array(x0, dtype=float), models=[], trials=[])
        self.solves.append(self.cur)


def trace_case(m, rec, poses, projs, init, max_nfev):
    """The two stage functions as PoseSolver.solve calls them (inverse_kinematics.py:404-406), traced."""
    skel = m.ik.load_skeleton()
    PoseSolver = sys.modules["inverse_kinematics_pino"].PoseSolver
    ps = PoseSolver(skel, None, [np.array(p) for p in poses], [np.array(p) for p in projs], obs_kps_format=m.pose_def.KpsFormat.COCO)
    init_param = m.ik.PoseShapeParam(np.array(init[:3]), np.array(init[3:57]).reshape(18, 3), np.array(init[57:]))
    results = []
    orig = m.ik.least_squares

    def wrapped(fun, x0, **kw):
        rec.begin(x0)
        r = orig(fun, x0, **kw)
        results.append(r)
        return r

    m.ik.least_squares = wrapped
    try:
        p1 = m.ik.solve_pose_reproj(ps.skel, np.array(ps.cam_poses_2d), ps.obs_kps_idxs, ps.cam_projs, ps.skel_kps_idxs, init_param, max_nfev)
        m.ik.solve_pose_bone_lens_reproj(ps.skel, np.array(ps.cam_poses_2d), ps.obs_kps_idxs, ps.cam_projs, ps.skel_kps_idxs, p1, max_nfev)
    finally:
        m.ik.least_squares = orig
    return results


def main():
    m = ref_shim.load_modules()
    g = np.load(f"{OUT}/ik_cases.npz")
    gc = np.load(f"{OUT}/ik_converged.npz")
    cases = []
    for i in range(len(g["frame"])):
        v = int(g["n_views"][i])
        init = np.concatenate([g["s1_x0"][i], g["s2_x0"][i][57:]])      # the start point PoseSolver.solve built (cold: DLT root)
        cases.append(dict(src=0, poses=g["poses"][i, :v], projs=g["projs"][i, :v], init=init, cold=bool(g["cold"][i]),
                          check=(g["s1_x"][i], g["s2_x"][i], int(g["s1_nfev"][i]), int(g["s2_nfev"][i]))))
    syn = np.flatnonzero(gc["source"] == 1)
    warm = [i for i in syn if gc["warm_init"][i]][:N_SYNTH_WARM]
    cold = [i for i in syn if not gc["warm_init"][i]][:N_SYNTH_COLD]
    for i, is_cold in [(i, True) for i in cold] + [(i, False) for i in warm]:
        v = int(gc["n_views"][i])
        cases.append(dict(src=1, poses=gc["poses"][i, :v], projs=gc["projs"][i, :v], init=gc["init"][i], cold=is_cold, check=None))

    n = len(cases)
    d = dict(case_poses=np.zeros((n, V_PAD, 17, 3)), case_projs=np.zeros((n, V_PAD, 3, 4)), case_nviews=np.zeros(n, np.int32),
             case_cold=np.zeros(n, bool), case_init=np.zeros((n, 68)), case_source=np.zeros(n, np.int32),
             case_x=np.zeros((n, 2, 68)), case_cost=np.zeros((n, 2)), case_nfev=np.zeros((n, 2), np.int32),
             case_njev=np.zeros((n, 2), np.int32), case_status=np.zeros((n, 2), np.int32))
    T = {k: [] for k in ("case", "stage", "model", "nfev", "x", "cost", "g", "s", "Delta", "alpha_in", "step", "alpha", "niter", "pred",
                         "cost_new", "actual", "ratio", "Delta_new", "status", "accepted")}
    jac_trial, jacs = [], []

    def pad(v):
        out = np.zeros(68)
        out[:len(v)] = v
        return out

    with TrfRecorder() as rec:
        for ci, c in enumerate(cases):
            n0 = len(rec.solves)
            res = trace_case(m, rec, c["poses"], c["projs"], c["init"], 50 if c["cold"] else 5)
            assert len(rec.solves) - n0 == 2
            v = len(c["poses"])
            d["case_poses"][ci, :v], d["case_projs"][ci, :v], d["case_nviews"][ci] = c["poses"], c["projs"], v
            d["case_cold"][ci], d["case_init"][ci], d["case_source"][ci] = c["cold"], c["init"], c["src"]
            if c["check"] is not None:   # instrumentation must not move a bit
                assert np.array_equal(res[0].x, c["check"][0]) and np.array_equal(res[1].x, c["check"][1]), ci
                assert (res[0].nfev, res[1].nfev) == c["check"][2:], ci
            for st, (r, sol) in enumerate(zip(res, rec.solves[n0:])):
                d["case_x"][ci, st] = pad(r.x)
                d["case_cost"][ci, st], d["case_nfev"][ci, st], d["case_njev"][ci, st], d["case_status"][ci, st] = r.cost, r.nfev, r.njev, r.status
                x = sol["x0"].copy()
                nfev = 1
                for ti, t in enumerate(sol["trials"]):
                    nfev += 1
                    if "actual" not in t:        # a non-finite residual (trf.py:505-507): never seen on these inputs
                        raise RuntimeError("non-finite trial")
                    if st == 0 and t["model"] == 0 and ti == 0:
                        jac_trial.append(len(T["case"]))
                        jacs.append(sol["models"][0]["J"])
                    acc = t["actual"] > 0
                    for k, val in dict(case=ci, stage=st, model=t["model"], nfev=nfev, x=pad(x), cost=t["cost"], g=pad(t["g"]),
                                       s=pad(sol["models"][t["model"]]["s"]), Delta=t["Delta"], alpha_in=t["alpha_in"], step=pad(t["step"]),
                                       alpha=t["alpha"], niter=t["niter"], pred=t["pred"], cost_new=t["cost"] - t["actual"],
                                       actual=t["actual"], ratio=t["ratio"], Delta_new=t["Delta_new"], status=t["status"],
                                       accepted=acc).items():
                        T[k].append(val)
                    if acc:
                        x = x + t["step"]      # trf.py:500,528: the same sum SciPy forms
                assert np.array_equal(x, r.x), (ci, st)
                assert nfev == r.nfev
            print(f"case {ci}: {'cold' if c['cold'] else 'warm'} {v} views, nfev {res[0].nfev}+{res[1].nfev}, "
                  f"status {res[0].status}/{res[1].status}", flush=True)
    for k, v in T.items():
        d["t_" + k] = np.array(v)
    m_max = max(j.shape[0] for j in jacs)
    J = np.zeros((len(jacs), m_max, 57))
    for i, j in enumerate(jacs):
        J[i, :j.shape[0]] = j
    d["jac_trial"], d["jac"] = np.array(jac_trial), J
    np.savez_compressed(f"{OUT}/ik_trf_traces.npz", **d)
    print("cases", n, "cold", int(d["case_cold"].sum()), "trials", len(T["case"]), "accepted", int(np.sum(T["accepted"])))


if __name__ == "__main__":
    main()
