"""CPU restatement of the trust-region solver the reference reaches through
``scipy.optimize.least_squares(fun, x0, max_nfev=k)`` (inverse_kinematics.py:236,
:274; mv_math_util.py:203), plus the analytic IK Jacobian.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE (see oracle/oracle_np.py header).

SciPy defaults at those call sites: method='trf', no bounds, tr_solver='exact',
jac='2-point', x_scale=1, ftol=xtol=gtol=1e-8, loss='linear'.  The functions
below restate, in plain NumPy and without importing SciPy:

  * ``fd_jacobian``      scipy/optimize/_numdiff.py: approx_derivative '2-point'
                         (h = sqrt(eps) * sign(x) * max(1, |x|), forward)
  * ``solve_tr_svd``     scipy/optimize/_lsq/common.py:57-168 solve_lsq_trust_region
  * ``trf``              scipy/optimize/_lsq/trf.py:401-560 trf_no_bounds
  * ``solve_tr_normal``  the same sub-problem from (J^T J, J^T f) only -- the
                         form the HIP kernel uses
  * ``solve_tr_normal_clean``  the same without LAPACK's rounding noise in the
                         numerically-null directions: ``trf(..., solver="ne_clean")`` is
                         the DETERMINISTIC whole-solve oracle of the truncated solves

tests/test_trf_restatement.py checks ``trf`` step for step against SciPy.
"""
from __future__ import annotations

import numpy as np

import oracle_np as o

EPS = np.finfo(float).eps


def fd_jacobian(fun, x, f0):
    """2-point forward differences exactly as SciPy's approx_derivative does them
    for an unbounded problem (_numdiff.py: _compute_absolute_step + _dense_difference)."""
    n = len(x)
    sign_x = (x >= 0).astype(float) * 2 - 1
    h = EPS ** 0.5 * sign_x * np.maximum(1.0, np.abs(x))
    J = np.empty((len(f0), n))
    for i in range(n):
        xp = x.copy()
        xp[i] = x[i] + h[i]
        dx = xp[i] - x[i]
        J[:, i] = (fun(xp) - f0) / dx
    return J


def solve_tr_svd(n, m, uf, s, V, Delta, initial_alpha=None, rtol=0.01, max_iter=10):
    """common.py:57-168.  Returns (p, alpha, n_iter)."""
    def phi_and_derivative(alpha):
        denom = s ** 2 + alpha
        p_norm = np.linalg.norm(suf / denom)
        return p_norm - Delta, -np.sum(suf ** 2 / denom ** 3) / p_norm

    suf = s * uf
    full_rank = (m >= n) and (s[-1] > EPS * m * s[0])
    if full_rank:
        p = -V.dot(uf / s)
        if np.linalg.norm(p) <= Delta:
            return p, 0.0, 0
    alpha_upper = np.linalg.norm(suf) / Delta
    if full_rank:
        phi, phi_prime = phi_and_derivative(0.0)
        alpha_lower = -phi / phi_prime
    else:
        alpha_lower = 0.0
    if initial_alpha is None or (not full_rank and initial_alpha == 0):
        alpha = max(0.001 * alpha_upper, (alpha_lower * alpha_upper) ** 0.5)
    else:
        alpha = initial_alpha
    it = -1
    for it in range(max_iter):
        if alpha < alpha_lower or alpha > alpha_upper:
            alpha = max(0.001 * alpha_upper, (alpha_lower * alpha_upper) ** 0.5)
        phi, phi_prime = phi_and_derivative(alpha)
        if phi < 0:
            alpha_upper = alpha
        ratio = phi / phi_prime
        alpha_lower = max(alpha_lower, alpha - ratio)
        alpha -= (phi + Delta) * ratio / Delta
        if abs(phi) < rtol * Delta:
            break
    p = -V.dot(suf / (s ** 2 + alpha))
    p *= Delta / np.linalg.norm(p)
    return p, alpha, it + 1


def trf(fun, jac, x0, max_nfev, ftol=1e-8, xtol=1e-8, gtol=1e-8, solver="svd", trace=None):
    """trf.py:401-560 (trf_no_bounds, x_scale = 1, linear loss).
    ``jac(x, f)`` returns the dense Jacobian.  Returns dict(x, cost, nfev, njev, status)."""
    x = np.array(x0, dtype=float)
    f = fun(x)
    nfev, njev = 1, 1
    J = jac(x, f)
    m, n = J.shape
    cost = 0.5 * np.dot(f, f)
    g = J.T.dot(f)
    Delta = np.linalg.norm(x)
    if Delta == 0:
        Delta = 1.0
    alpha = 0.0
    status = None
    while True:
        if np.linalg.norm(g, ord=np.inf) < gtol:
            status = 1
        if status is not None or nfev == max_nfev:
            break
        if solver == "svd":
            U, s, Vt = np.linalg.svd(J, full_matrices=False)
            V = Vt.T
            uf = U.T.dot(f)
        else:
            lam, V, = _eigh_desc(J.T.dot(J))
        if trace is not None and solver == "ne_clean":
            trace.append(dict(model=True, nfev=nfev, lam=lam.copy(), g_inf=np.linalg.norm(g, ord=np.inf)))
        actual_reduction = -1.0
        x_new = f_new = cost_new = None
        while actual_reduction <= 0 and nfev < max_nfev:
            predicted_reduction = None
            alpha_in = alpha
            if solver == "svd":
                step, alpha, _ = solve_tr_svd(n, m, uf, s, V, Delta, initial_alpha=alpha)
            elif solver == "ne_clean":
                step, alpha, predicted_reduction, step_norm_clean = solve_tr_normal_clean(lam, V, g, Delta, initial_alpha=alpha)
            else:
                step, alpha, _ = solve_tr_normal(n, m, lam, V, g, Delta, initial_alpha=alpha)
            if predicted_reduction is None:
                Js = J.dot(step)
                predicted_reduction = -(0.5 * np.dot(Js, Js) + np.dot(step, g))
            x_new = x + step
            f_new = fun(x_new)
            nfev += 1
            # (the noise-free form accounts for the step as SciPy does for a rank-deficient model: |p| = Delta, the share of it that
            # the reference spends on numerically-null directions modelled by the absorber and not taken)
            step_norm = step_norm_clean if solver == "ne_clean" else np.linalg.norm(step)
            if not np.all(np.isfinite(f_new)):
                Delta = 0.25 * step_norm
                continue
            cost_new = 0.5 * np.dot(f_new, f_new)
            actual_reduction = cost - cost_new
            # update_tr_radius, common.py:222-245
            if predicted_reduction > 0:
                ratio = actual_reduction / predicted_reduction
            elif predicted_reduction == actual_reduction == 0:
                ratio = 1
            else:
                ratio = 0
            Delta_new = Delta
            if ratio < 0.25:
                Delta_new = 0.25 * step_norm
            elif ratio > 0.75 and step_norm > 0.95 * Delta:
                Delta_new = Delta * 2.0
            if trace is not None:
                trace.append(dict(nfev=nfev, alpha=alpha, Delta=Delta, step_norm=step_norm, cost_new=cost_new,
                                  ratio=ratio, accepted=actual_reduction > 0, x=x.copy(), alpha_in=alpha_in, Delta_new=Delta_new,
                                  pred=predicted_reduction, cost=cost, step=step.copy()))
            # check_termination, common.py:705-717
            ftol_ok = actual_reduction < ftol * cost and ratio > 0.25
            xtol_ok = step_norm < xtol * (xtol + np.linalg.norm(x))
            status = 4 if (ftol_ok and xtol_ok) else 2 if ftol_ok else 3 if xtol_ok else None
            if status is not None:
                break
            alpha *= Delta / Delta_new
            Delta = Delta_new
        if actual_reduction > 0:
            x, f, cost = x_new, f_new, cost_new
            J = jac(x, f)
            njev += 1
            g = J.T.dot(f)
    if status is None:
        status = 0
    return dict(x=x, cost=cost, nfev=nfev, njev=njev, status=status, fun=f, grad=g)


def solve_tr_normal_clean(lam, V, g, Delta, initial_alpha=0.0, null_tol=1e-13, rtol=0.01, max_iter=10):
    """solve_lsq_trust_region's rank-deficient branch (common.py:116-168) WITHOUT the rounding noise SciPy's LAPACK leaves in the
    numerically-null singular directions -- the deterministic form of the reference's step, and the production IK kernel's
    ("the absorber"; csrc/mvmc_eigh_tri.h tr_solve_tri / tr_solve_eig_w1):

      * the null cluster (lam <= null_tol * lam_max) carries no step;
      * ONE virtual direction (lam = 0, suf = 1e-8 |g|) stands for what those directions hold in the reference (finite-difference
        noise s u^T f ~ 1e-8 |g|): it takes part in phi(alpha), in the normalisation |p| = Delta and in the predicted reduction
        exactly as a singular triplet would, and its coefficient is not applied to x.

    lam, V: eigen-decomposition of J^T J, descending (``_eigh_desc``).  Returns (step, alpha, predicted_reduction, |p| = Delta)."""
    keep = lam > null_tol * lam[0]
    lam_k, V_k = lam[keep], V[:, keep]
    suf = np.append(V_k.T.dot(g), 1e-8 * np.linalg.norm(g))
    lam_a = np.append(lam_k, 0.0)
    alpha_upper, alpha_lower = np.linalg.norm(suf) / Delta, 0.0
    alpha = max(0.001 * alpha_upper, 0.0) if (initial_alpha is None or initial_alpha == 0) else initial_alpha
    for _ in range(max_iter):
        if alpha < alpha_lower or alpha > alpha_upper:
            alpha = max(0.001 * alpha_upper, (alpha_lower * alpha_upper) ** 0.5)
        den = lam_a + alpha
        p_norm = np.linalg.norm(suf / den)
        phi, phi_prime = p_norm - Delta, -np.sum(suf ** 2 / den ** 3) / p_norm
        if phi < 0:
            alpha_upper = alpha
        ratio = phi / phi_prime
        alpha_lower = max(alpha_lower, alpha - ratio)
        alpha -= (phi + Delta) * ratio / Delta
        if abs(phi) < rtol * Delta:
            break
    c = -suf / (lam_a + alpha)
    c *= Delta / np.linalg.norm(c)
    pred = -(0.5 * np.sum(lam_a * c * c) + np.sum(suf * c))
    return V_k.dot(c[:-1]), alpha, pred, float(np.linalg.norm(c))


def _eigh_desc(A):
    lam, V = np.linalg.eigh(A)
    return np.maximum(lam[::-1], 0.0), V[:, ::-1]


def solve_tr_normal(n, m, lam, V, g, Delta, initial_alpha=None, rtol=0.01, max_iter=10):
    """solve_lsq_trust_region from the normal equations: with J^T J = V diag(lam) V^T,
    s^2 = lam and s * (U^T f) = V^T g, so p(alpha) = -V (V^T g / (lam + alpha))."""
    s = np.sqrt(lam)
    suf = V.T.dot(g)
    # uf = suf / s where s > 0 (only needed for the full-rank Gauss-Newton shortcut)
    full_rank = (m >= n) and (s[-1] > EPS * m * s[0])
    uf = np.divide(suf, s, out=np.zeros_like(suf), where=s > 0)
    return solve_tr_svd(n, m, uf, s, V, Delta, initial_alpha=initial_alpha, rtol=rtol, max_iter=max_iter) \
        if full_rank else _solve_tr_deficient(suf, lam, V, Delta, initial_alpha, rtol, max_iter)


def _solve_tr_deficient(suf, lam, V, Delta, initial_alpha, rtol, max_iter):
    def phi_and_derivative(alpha):
        denom = lam + alpha
        p_norm = np.linalg.norm(suf / denom)
        return p_norm - Delta, -np.sum(suf ** 2 / denom ** 3) / p_norm

    alpha_upper = np.linalg.norm(suf) / Delta
    alpha_lower = 0.0
    if initial_alpha is None or initial_alpha == 0:
        alpha = max(0.001 * alpha_upper, 0.0)
    else:
        alpha = initial_alpha
    it = -1
    for it in range(max_iter):
        if alpha < alpha_lower or alpha > alpha_upper:
            alpha = max(0.001 * alpha_upper, (alpha_lower * alpha_upper) ** 0.5)
        phi, phi_prime = phi_and_derivative(alpha)
        if phi < 0:
            alpha_upper = alpha
        ratio = phi / phi_prime
        alpha_lower = max(alpha_lower, alpha - ratio)
        alpha -= (phi + Delta) * ratio / Delta
        if abs(phi) < rtol * Delta:
            break
    p = -V.dot(suf / (lam + alpha))
    p *= Delta / np.linalg.norm(p)
    return p, alpha, it + 1


# ----------------------------------------------------------------------------
# analytic Jacobian of the IK residual
# ----------------------------------------------------------------------------
def _axis_rot(c, a):
    cs, sn = np.cos(a), np.sin(a)
    if c == 0:
        return np.array([[1, 0, 0], [0, cs, -sn], [0, sn, cs]])
    if c == 1:
        return np.array([[cs, 0, sn], [0, 1, 0], [-sn, 0, cs]])
    return np.array([[cs, -sn, 0], [sn, cs, 0], [0, 0, 1]])


def ik_jacobian(root, euler, side_blens, obs, projs, with_blens):
    """d residual / d [root(3), euler(54), (side_blens(11))] -> (V*16*2, 57|68).
    Residual as oracle_np.ik_residual (inverse_kinematics.py:219-234)."""
    bone_dirs, _ = o.skeleton_constants()
    euler = np.asarray(euler, float).reshape(18, 3)
    pos, G = o.forward_kinematics(root, euler, side_blens, bone_dirs)
    Rg = G[:, :3, :3]
    par = o.SKEL_PARENTS
    n = 57 + (11 if with_blens else 0)
    # rotation axes in the global frame: R_a = Rx Ry Rz inside parent frame Rg_p
    axes = np.zeros((18, 3, 3))
    for a in range(18):
        Rp = Rg[par[a]] if par[a] >= 0 else np.eye(3)
        Rx = _axis_rot(0, euler[a, 0])
        Ry = _axis_rot(1, euler[a, 1])
        axes[a, 0] = Rp @ np.array([1.0, 0, 0])
        axes[a, 1] = Rp @ Rx @ np.array([0, 1.0, 0])
        axes[a, 2] = Rp @ Rx @ Ry @ np.array([0, 0, 1.0])
    dX = np.zeros((16, 3, n))
    for r, k in enumerate(o.IK_SKEL_IDX):
        dX[r, :, 0:3] = np.eye(3)
        j = k
        while j != 0:
            p = par[j]
            # bone j hangs off parent p: length derivative
            if with_blens:
                dX[r, :, 57 + o.SIDE_TO_FULL[j]] += Rg[p] @ bone_dirs[j]
            # every strict ancestor's angles move k
            for c in range(3):
                dX[r, :, 3 + 3 * p + c] = np.cross(axes[p, c], pos[k] - pos[p])
            j = p
    V = len(projs)
    J = np.zeros((V, 16, 2, n))
    X = pos[o.IK_SKEL_IDX]
    for v in range(V):
        P = projs[v]
        h = X @ P[:, :3].T + P[:, 3]
        w = h[:, 2] + 1e-5
        for r in range(16):
            du = (P[0, :3] - h[r, 0] / w[r] * P[2, :3]) / w[r]
            dv = (P[1, :3] - h[r, 1] / w[r] * P[2, :3]) / w[r]
            J[v, r, 0] = obs[v, r, 2] * (du @ dX[r])
            J[v, r, 1] = obs[v, r, 2] * (dv @ dX[r])
    return J.reshape(V * 32, n)


def pose_solver_solve_clean(cam_poses_2d, cam_projs, init=None, return_info=False):
    """PoseSolver(...).solve() (inverse_kinematics.py:351-433) with both least_squares calls replaced by the NOISE-FREE restatement of the
    same algorithm (``trf(solver="ne_clean")``, analytic Jacobian): oracle_np.pose_solver_solve's cold start (DLT + the reference's
    post-optimisation, root = midpoint of the hips, zero angles, reference lengths; budget 50) and warm start (budget 5), deterministic
    where SciPy's result depends on LAPACK's rounding.  The per-solve oracle of tests/test_gpu_ik_whole_solves.py as a drop-in for
    tracker_np.OracleTracker(solver=...): the whole-sequence oracle of tests/test_gpu_tracker.py."""
    projs = np.asarray(cam_projs, np.float64)
    poses18 = [o.add_mid_spine(p) for p in cam_poses_2d]
    obs = np.array(poses18)[:, o.IK_OBS_IDX, :]
    bone_dirs, ref_side = o.skeleton_constants()
    if init is None:
        p3d = o.triangulate_groups(projs, poses18, 0.01, True)
        root = 0.5 * (p3d[o.COCO_L_HIP, :3] + p3d[o.COCO_R_HIP, :3])
        euler = np.zeros((o.N_SKEL, 3))
        blens = ref_side.copy()
        nfev = 50
    else:
        root, euler, blens = (np.asarray(a, np.float64) for a in init)
        nfev = 5
    side0 = np.asarray(blens, np.float64).copy()
    f1 = lambda x: o.ik_residual(x[:3], x[3:57], side0, obs, projs, bone_dirs)
    j1 = lambda x, f: ik_jacobian(x[:3], x[3:57], side0, obs, projs, False)
    f2 = lambda x: o.ik_residual(x[:3], x[3:57], x[57:], obs, projs, bone_dirs)
    j2 = lambda x, f: ik_jacobian(x[:3], x[3:57], x[57:], obs, projs, True)
    r1 = trf(f1, j1, np.concatenate([np.ravel(root), np.ravel(euler)]), nfev, solver="ne_clean")
    r2 = trf(f2, j2, np.concatenate([r1["x"], side0]), nfev, solver="ne_clean")
    x = r2["x"]
    joints, _ = o.forward_kinematics(x[:3], x[3:57], x[57:], bone_dirs)
    out = (x[:3].copy(), x[3:57].reshape(-1, 3).copy(), x[57:].copy())
    if return_info:
        return out, joints, dict(res1=r1, res2=r2)
    return out, joints
