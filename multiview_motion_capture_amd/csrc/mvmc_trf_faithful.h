// The reference's least-squares solver restated literally, as plain single-thread C++ that compiles for the host and for the device.
//
//   scipy.optimize.least_squares(fun, x0, max_nfev=k) with SciPy's defaults -- method 'trf' without bounds, tr_solver 'exact' (an SVD
//   of the Jacobian), jac '2-point' (forward differences, step sqrt(eps) * sign(x) * max(1, |x|)), x_scale 1, ftol = xtol = gtol = 1e-8,
//   linear loss -- as the reference calls it for the IK stages (inverse_kinematics.py:236, :274) and for the post-optimisation of the
//   triangulation (mv_math_util.py:203).  SciPy sources followed: _numdiff.py (approx_derivative, _dense_difference),
//   _lsq/trf.py:401-560 (trf_no_bounds), _lsq/common.py:57-168 (solve_lsq_trust_region), :222-245 (update_tr_radius), :705-717
//   (check_termination).
//
// Two users, which is why it is a header of host/device functions and why nothing in it is tuned:
//   * the device's TRF-faithful IK solver (mvmc_debug_ik_solve_fd, mvmc_ik_fd.hip): one wave per solve, a diagnostic that answers
//     "how far from the reference is the production solver's analytic-Jacobian / Krylov step, and how far is the reference's own
//     algorithm when only the rounding differs" -- never on the hot path;
//   * the C++ CPU baseline (oracle/cpu_twin), the second CPU restatement of the path that bench.py times next to the NumPy one.
//
// The SVD is a Householder QR followed by one-sided (Hestenes) Jacobi on R, which resolves small singular values to high RELATIVE
// accuracy: the directions SciPy's LAPACK (gesdd) reports with s ~ 1e-8 s_max -- finite-difference noise along the gauge directions of
// the skeleton -- are the ones that decide where a rank-deficient trust-region step goes (DESIGN.md "IK parity").
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define MVMC_HD __host__ __device__
#else
#define MVMC_HD
#endif

namespace trf_faithful {

constexpr double kEps = 2.220446049250313e-16;
constexpr int kMaxParams = 72;   // private copy of x inside the finite-difference loop

// Execution policy.  The algorithms below are written once as loops "for (i = ex.lane(); i < n; i += ex.lanes())" with sums over the
// lanes and a sync after every phase whose results other lanes read:
//   Serial   the host (and any single thread): one lane, nothing to reduce or order;
//   Wave64   the device: the 64 lanes of a one-wave workgroup share the vectors and matrices of a solve in global memory.
// Every branch is taken on reduced (hence lane-identical) values, so the lanes of a wave stay together.
struct Serial {
    MVMC_HD int lane() const { return 0; }
    MVMC_HD int lanes() const { return 1; }
    MVMC_HD double sum(double v) const { return v; }
    MVMC_HD void sync() const {}
};
#if defined(__HIPCC__)
struct Wave64 {
    __device__ int lane() const { return threadIdx.x & 63; }
    __device__ int lanes() const { return 64; }
    __device__ double sum(double v) const {
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);   // butterfly: bitwise the same in every lane
        return v;
    }
    __device__ void sync() const { __syncthreads(); }   // one wave per workgroup: orders its global-memory traffic
};
#endif

template <class Ex>
MVMC_HD inline double dot(const Ex& ex, const double* a, const double* b, int n) {
    double s = 0.0;
    for (int i = ex.lane(); i < n; i += ex.lanes()) s += a[i] * b[i];
    return ex.sum(s);
}
template <class Ex>
MVMC_HD inline double norm2(const Ex& ex, const double* v, int n) { return sqrt(dot(ex, v, v, n)); }

// approx_derivative(fun, x0, method='2-point', f0=f0) for an unbounded problem: J (m x n, column-major with leading dimension ld).
// A lane evaluates whole columns (fun is called concurrently with different outputs).
template <class Ex, class Fun>
MVMC_HD void fd_jacobian(const Ex& ex, const Fun& fun, int n, int m, const double* x, const double* f0, double* J, int ld) {
    const double rel = sqrt(kEps);
    for (int j = ex.lane(); j < n; j += ex.lanes()) {
        double xl[kMaxParams];
        for (int i = 0; i < n; ++i) xl[i] = x[i];
        const double x0 = xl[j];
        const double h = rel * (x0 >= 0.0 ? 1.0 : -1.0) * fmax(1.0, fabs(x0));
        xl[j] = x0 + h;
        const double dx = xl[j] - x0;
        double* col = J + (size_t)j * ld;
        fun(xl, col);
        for (int i = 0; i < m; ++i) col[i] = (col[i] - f0[i]) / dx;
    }
    ex.sync();
}

// Thin SVD pieces of the m x n matrix A (column-major, leading dimension ld >= max(m, n)) that solve_lsq_trust_region needs:
// s[k] singular values, V (n x n, column-major) right singular vectors, suf[k] = s_k * (u_k . f).  Not sorted.  A and fq (a copy of f,
// length >= max(m, n)) are destroyed; rows m..max(m, n)-1 of A are zeroed here (zero rows change neither s nor V).
template <class Ex>
MVMC_HD inline void svd_pieces(const Ex& ex, int m, int n, double* A, int ld, double* fq, double* s, double* suf, double* V) {
    const int mm = m > n ? m : n;
    if (mm > m) {
        for (int j = 0; j < n; ++j)
            for (int i = m + ex.lane(); i < mm; i += ex.lanes()) A[(size_t)j * ld + i] = 0.0;
        for (int i = m + ex.lane(); i < mm; i += ex.lanes()) fq[i] = 0.0;
        ex.sync();
    }
    // ---- Householder QR: A = Q R, fq <- Q^T f; R is left in the upper triangle of the first n rows ----
    for (int k = 0; k < n; ++k) {
        double* ak = A + (size_t)k * ld;
        const double sig = dot(ex, ak + k + 1, ak + k + 1, mm - k - 1);
        if (sig == 0.0) continue;
        const double alpha = ak[k];
        ex.sync();                                  // every lane has read alpha before lane 0 overwrites it
        const double nrm = sqrt(alpha * alpha + sig);
        const double beta = alpha >= 0.0 ? -nrm : nrm;
        const double tau = (beta - alpha) / beta;
        const double sc = 1.0 / (alpha - beta);
        for (int i = k + 1 + ex.lane(); i < mm; i += ex.lanes()) ak[i] *= sc;   // v = (1, ak[k+1..])
        if (ex.lane() == 0) ak[k] = beta;
        ex.sync();
        for (int j = k + 1; j <= n; ++j) {          // the columns behind k, then fq
            double* aj = j < n ? A + (size_t)j * ld : fq;
            const double head = aj[k];
            const double w = tau * (head + dot(ex, ak + k + 1, aj + k + 1, mm - k - 1));
            ex.sync();
            if (ex.lane() == 0) aj[k] = head - w;
            for (int i = k + 1 + ex.lane(); i < mm; i += ex.lanes()) aj[i] -= w * ak[i];
        }
        ex.sync();
    }
    for (int k = 0; k < n; ++k) {   // R as a dense n x n matrix, V = I
        double* ak = A + (size_t)k * ld;
        double* vk = V + (size_t)k * n;
        for (int i = ex.lane(); i < n; i += ex.lanes()) {
            if (i > k) ak[i] = 0.0;
            vk[i] = i == k ? 1.0 : 0.0;
        }
    }
    ex.sync();
    // ---- one-sided Jacobi on the columns of R: R V = (columns s_k u'_k); s holds the squared column norms meanwhile ----
    for (int k = 0; k < n; ++k) {
        const double v = dot(ex, A + (size_t)k * ld, A + (size_t)k * ld, n);
        if (ex.lane() == 0) s[k] = v;
    }
    ex.sync();
    for (int sweep = 0; sweep < 40; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < n - 1; ++p) {
            double* ap = A + (size_t)p * ld;
            double* vp = V + (size_t)p * n;
            for (int q = p + 1; q < n; ++q) {
                const double al = s[p], be = s[q];
                if (al == 0.0 || be == 0.0) continue;
                double* aq = A + (size_t)q * ld;
                const double ga = dot(ex, ap, aq, n);
                if (fabs(ga) <= 1e-15 * sqrt(al * be)) continue;
                ++rotated;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                double* vq = V + (size_t)q * n;
                ex.sync();                          // s[p], s[q] read by every lane before they change
                for (int i = ex.lane(); i < n; i += ex.lanes()) {
                    const double x = ap[i], y = aq[i];
                    ap[i] = c * x - sn * y; aq[i] = sn * x + c * y;
                    const double u = vp[i], w = vq[i];
                    vp[i] = c * u - sn * w; vq[i] = sn * u + c * w;
                }
                if (ex.lane() == 0) { s[p] = al - t * ga; s[q] = be + t * ga; }
                ex.sync();
            }
        }
        for (int k = 0; k < n; ++k) {               // refresh the norms (the running updates drift)
            const double v = dot(ex, A + (size_t)k * ld, A + (size_t)k * ld, n);
            ex.sync();
            if (ex.lane() == 0) s[k] = v;
        }
        ex.sync();
        if (!rotated) break;
    }
    for (int k = 0; k < n; ++k) {
        const double sf = dot(ex, A + (size_t)k * ld, fq, n);   // (s_k u'_k) . (Q^T f)
        const double sk = sqrt(s[k]);
        ex.sync();
        if (ex.lane() == 0) { suf[k] = sf; s[k] = sk; }
    }
    ex.sync();
    {   // columns that cancelled to the underflow range (|a_k| ~ 1e-150) are exact zeros that lost their last bits: no triplet
        double smax = 0.0;
        for (int k = 0; k < n; ++k) smax = fmax(smax, s[k]);
        ex.sync();
        for (int k = ex.lane(); k < n; k += ex.lanes())
            if (s[k] <= 1e-30 * smax) { s[k] = 0.0; suf[k] = 0.0; }
        ex.sync();
    }
    if (m < n) {
        // SciPy's thin SVD (full_matrices=False) has min(m, n) singular triplets: the n - m directions that the appended zero rows
        // leave at rounding level are not part of it (their noise projections would otherwise enter the step through s^2 + alpha ~ 0)
        for (int k = ex.lane(); k < n; k += ex.lanes()) {   // rank of s[k] in descending order; fq is free by now
            int above = 0;
            for (int j = 0; j < n; ++j) above += (s[j] > s[k]) || (s[j] == s[k] && j < k);
            fq[k] = above >= m ? 1.0 : 0.0;
        }
        ex.sync();
        for (int k = ex.lane(); k < n; k += ex.lanes())
            if (fq[k] != 0.0) { s[k] = 0.0; suf[k] = 0.0; }
        ex.sync();
    }
}

// solve_lsq_trust_region(n, m, uf, s, V, Delta, initial_alpha) (common.py:57-168) with uf = suf / s.  p <- step (n); returns alpha.
// The scalar iteration runs redundantly in every lane (identical inputs, identical results); the step is formed row-parallel.
template <class Ex>
MVMC_HD inline double solve_lsq_trust_region(const Ex& ex, int n, int m, const double* s, const double* suf, const double* V,
                                             double Delta, double initial_alpha, double* p, double* coef) {
    double smax = 0.0, smin = 1e300, sufn = 0.0;
    for (int k = 0; k < n; ++k) { smax = fmax(smax, s[k]); smin = fmin(smin, s[k]); sufn += suf[k] * suf[k]; }
    sufn = sqrt(sufn);
    const bool full_rank = (m >= n) && (smin > kEps * m * smax);
    auto form_step = [&](double alpha, bool gauss_newton) {
        ex.sync();
        for (int k = ex.lane(); k < n; k += ex.lanes())
            coef[k] = gauss_newton ? -suf[k] / (s[k] * s[k]) : (suf[k] == 0.0 ? 0.0 : -suf[k] / (s[k] * s[k] + alpha));
        ex.sync();
        for (int i = ex.lane(); i < n; i += ex.lanes()) {
            double acc = 0.0;
            for (int k = 0; k < n; ++k) acc += V[(size_t)k * n + i] * coef[k];
            p[i] = acc;
        }
        ex.sync();
        return norm2(ex, p, n);
    };
    if (full_rank) {
        if (form_step(0.0, true) <= Delta) return 0.0;
    }
    auto phi_and_derivative = [&](double alpha, double* phi_prime) {
        double pn = 0.0, d3 = 0.0;
        for (int k = 0; k < n; ++k) {
            if (suf[k] == 0.0) continue;    // 0 / (s^2 + alpha): the exactly-null columns contribute nothing for alpha > 0
            const double denom = s[k] * s[k] + alpha;
            const double t = suf[k] / denom;
            pn += t * t;
            d3 += suf[k] * suf[k] / (denom * denom * denom);
        }
        pn = sqrt(pn);
        *phi_prime = -d3 / pn;
        return pn - Delta;
    };
    double alpha_upper = sufn / Delta;
    double alpha_lower = 0.0;
    if (full_rank) {
        double pp;
        const double phi = phi_and_derivative(0.0, &pp);
        alpha_lower = -phi / pp;
    }
    double alpha;
    if (!full_rank && initial_alpha == 0.0) alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
    else alpha = initial_alpha;
    for (int it = 0; it < 10; ++it) {
        if (alpha < alpha_lower || alpha > alpha_upper) alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
        double pp;
        const double phi = phi_and_derivative(alpha, &pp);
        if (phi < 0.0) alpha_upper = alpha;
        const double ratio = phi / pp;
        alpha_lower = fmax(alpha_lower, alpha - ratio);
        alpha -= (phi + Delta) * ratio / Delta;
        if (fabs(phi) < 0.01 * Delta) break;
    }
    // When the Gauss-Newton step of a rank-deficient problem is shorter than Delta, phi(alpha) < 0 for every alpha > 0: the iteration
    // above drives alpha to zero (the resets divide it by 1000 each time) and its LAST Newton update, which nothing checks, can land
    // below zero.  SciPy's LAPACK SVD reports exactly-null directions as noise triplets (s ~ 1e-17 s_max) whose 1 / (s^2 + alpha)
    // makes phi cross zero at alpha ~ 1e-20 and so stops that last update at ~ -1e-20; the Jacobi SVD above returns exact zeros
    // there, and the same update would end at a negative alpha of ordinary size.  Clamping it restores SciPy's outcome: the step
    // is the minimum-norm Gauss-Newton step stretched to |p| = Delta.
    // That is the m >= n case.  With FEWER residuals than unknowns (the triangulation post-optimisation of a two-view cluster: 36 x 54)
    // SciPy's thin SVD holds null triplets only where J has dependent ROWS (a joint that a view scores 0: the factorisation above
    // leaves them at rounding level, like LAPACK, and they stop the update by themselves); with independent rows nothing stops it, and the
    // reference really takes -V (suf / (s^2 + alpha)) with that negative alpha of ordinary size (denominators of both signs),
    // stretched to Delta: kept as it is.
    // (Which of the two a build sees for a zero row -- an exact zero or a triplet at rounding level -- depends on its contraction of
    // multiply-adds; the rule covers both: clamp where the thin SVD has an exactly-null triplet, rank < min(m, n).)
    int rank = 0;
    for (int k = 0; k < n; ++k) rank += s[k] > 0.0 ? 1 : 0;
    if (!full_rank && alpha < 0.0 && rank < (m < n ? m : n)) alpha = 0.0;
    const double pn = form_step(alpha, false);
    const double sc = Delta / pn;
    for (int i = ex.lane(); i < n; i += ex.lanes()) p[i] *= sc;
    ex.sync();
    return alpha;
}

struct Result {
    double cost;
    int nfev, njev, status;
};

// Doubles of workspace trf() needs for an m x n problem
MVMC_HD inline size_t work_doubles(int m, int n) {
    const size_t mm = (size_t)(m > n ? m : n);
    return 2 * mm * n + (size_t)n * n + 3 * mm + 6 * (size_t)n;
}

// trf_no_bounds (trf.py:401-560): x (n) in/out, shared by the lanes; fun(x, f) writes the m residuals.
template <class Ex, class Fun>
MVMC_HD Result trf(const Ex& ex, const Fun& fun, int n, int m, double* x, int max_nfev, double* work, double ftol = 1e-8,
                   double xtol = 1e-8, double gtol = 1e-8) {
    const int mm = m > n ? m : n;
    double* J = work;                       // mm x n, kept for J^T f and the predicted reduction
    double* A = J + (size_t)mm * n;         // mm x n, destroyed by the SVD
    double* V = A + (size_t)mm * n;         // n x n
    double* f = V + (size_t)n * n;          // mm
    double* f_new = f + mm;                 // mm
    double* fq = f_new + mm;                // mm
    double* g = fq + mm;                    // n each from here
    double* s = g + n;
    double* suf = s + n;
    double* step = suf + n;
    double* x_new = step + n;
    double* coef = x_new + n;
    auto evaluate = [&](const double* xx, double* ff) {   // one lane evaluates, everybody reads
        ex.sync();
        if (ex.lane() == 0) fun(xx, ff);
        ex.sync();
    };
    auto gradient = [&]() {
        for (int j = 0; j < n; ++j) {
            const double v = dot(ex, J + (size_t)j * mm, f, m);
            if (ex.lane() == 0) g[j] = v;
        }
        ex.sync();
    };
    evaluate(x, f);
    int nfev = 1, njev = 1;
    fd_jacobian(ex, fun, n, m, x, f, J, mm);
    double cost = 0.5 * dot(ex, f, f, m);
    gradient();
    double Delta = norm2(ex, x, n);
    if (Delta == 0.0) Delta = 1.0;
    double alpha = 0.0;
    int status = -1;
    while (true) {
        double g_norm = 0.0;
        for (int j = 0; j < n; ++j) g_norm = fmax(g_norm, fabs(g[j]));
        if (g_norm < gtol) status = 1;
        if (status != -1 || nfev == max_nfev) break;
        for (int j = 0; j < n; ++j)
            for (int i = ex.lane(); i < m; i += ex.lanes()) A[(size_t)j * mm + i] = J[(size_t)j * mm + i];
        for (int i = ex.lane(); i < m; i += ex.lanes()) fq[i] = f[i];
        ex.sync();
        svd_pieces(ex, m, n, A, mm, fq, s, suf, V);
        double actual = -1.0, cost_new = cost;
        while (actual <= 0.0 && nfev < max_nfev) {
            alpha = solve_lsq_trust_region(ex, n, m, s, suf, V, Delta, alpha, step, coef);
            // predicted_reduction = -evaluate_quadratic(J, g, step)
            double q = 0.0;
            for (int i = ex.lane(); i < m; i += ex.lanes()) {
                double js = 0.0;
                for (int j = 0; j < n; ++j) js += J[(size_t)j * mm + i] * step[j];
                q += js * js;
            }
            q = ex.sum(q);
            const double pred = -(0.5 * q + dot(ex, step, g, n));
            for (int j = ex.lane(); j < n; j += ex.lanes()) x_new[j] = x[j] + step[j];
            evaluate(x_new, f_new);
            ++nfev;
            const double step_norm = norm2(ex, step, n);
            cost_new = 0.5 * dot(ex, f_new, f_new, m);
            if (!isfinite(cost_new)) { Delta = 0.25 * step_norm; continue; }
            actual = cost - cost_new;
            double ratio;
            if (pred > 0.0) ratio = actual / pred;
            else if (pred == 0.0 && actual == 0.0) ratio = 1.0;
            else ratio = 0.0;
            double Delta_new = Delta;
            if (ratio < 0.25) Delta_new = 0.25 * step_norm;
            else if (ratio > 0.75 && step_norm > 0.95 * Delta) Delta_new = Delta * 2.0;
            const bool f_ok = (actual < ftol * cost) && (ratio > 0.25);
            const bool x_ok = step_norm < xtol * (xtol + norm2(ex, x, n));
            if (f_ok && x_ok) status = 4; else if (f_ok) status = 2; else if (x_ok) status = 3;
            if (status != -1) break;
            alpha *= Delta / Delta_new;
            Delta = Delta_new;
        }
        if (actual > 0.0) {
            ex.sync();
            for (int j = ex.lane(); j < n; j += ex.lanes()) x[j] = x_new[j];
            for (int i = ex.lane(); i < m; i += ex.lanes()) f[i] = f_new[i];
            ex.sync();
            cost = cost_new;
            fd_jacobian(ex, fun, n, m, x, f, J, mm);
            ++njev;
            gradient();
        }
    }
    if (status == -1) status = 0;
    Result r;
    r.cost = cost; r.nfev = nfev; r.njev = njev; r.status = status;
    return r;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The IK residual of the reference (inverse_kinematics.py:219-234, :258-272) on top of foward_kinematics (:176-199) with
// Quaternions.from_euler / transforms (Quaternions.py:449-462, :335-366).
// ---------------------------------------------------------------------------------------------------------------------------------
struct Skel {
    double dirs[18][3];
    int parents[18];
    int side_map[18];
    int n_side;
};

MVMC_HD inline void quat_mul(const double* q, const double* r, double* o) {
    o[0] = r[0] * q[0] - r[1] * q[1] - r[2] * q[2] - r[3] * q[3];
    o[1] = r[0] * q[1] + r[1] * q[0] - r[2] * q[3] + r[3] * q[2];
    o[2] = r[0] * q[2] + r[1] * q[3] + r[2] * q[0] - r[3] * q[1];
    o[3] = r[0] * q[3] - r[1] * q[2] + r[2] * q[1] + r[3] * q[0];
}

MVMC_HD inline void euler_to_rot(const double* e, double* R) {
    const double inv = 1.0 / (1.0 + 1e-10);
    const double sx = sin(e[0] / 2.0), cx = cos(e[0] / 2.0), sy = sin(e[1] / 2.0), cy = cos(e[1] / 2.0);
    const double sz = sin(e[2] / 2.0), cz = cos(e[2] / 2.0);
    const double q0[4] = {cx, inv * sx, 0.0, 0.0};
    const double q1[4] = {cy, 0.0, inv * sy, 0.0};
    const double q2[4] = {cz, 0.0, 0.0, inv * sz};
    double q12[4], q[4];
    quat_mul(q1, q2, q12);
    quat_mul(q0, q12, q);
    const double qw = q[0], qx = q[1], qy = q[2], qz = q[3];
    const double x2 = qx + qx, y2 = qy + qy, z2 = qz + qz;
    const double xx = qx * x2, yy = qy * y2, wx = qw * x2;
    const double xy = qx * y2, yz = qy * z2, wy = qw * y2;
    const double xz = qx * z2, zz = qz * z2, wz = qw * z2;
    R[0] = 1.0 - (yy + zz); R[1] = xy - wz; R[2] = xz + wy;
    R[3] = xy + wz; R[4] = 1.0 - (xx + zz); R[5] = yz - wx;
    R[6] = xz - wy; R[7] = yz + wx; R[8] = 1.0 - (xx + yy);
}

// pos (18,3) from root (3), euler (18,3), side lengths (n_side)
MVMC_HD inline void forward_kinematics(const Skel& sk, const double* root, const double* euler, const double* side, double* pos) {
    double Rg[18][9];
    euler_to_rot(euler, Rg[0]);
    pos[0] = root[0]; pos[1] = root[1]; pos[2] = root[2];
    for (int j = 1; j < 18; ++j) {
        const int p = sk.parents[j];
        const double len = side[sk.side_map[j]];
        const double o0 = sk.dirs[j][0] * len, o1 = sk.dirs[j][1] * len, o2 = sk.dirs[j][2] * len;
        const double* G = Rg[p];
        pos[j * 3] = G[0] * o0 + G[1] * o1 + G[2] * o2 + pos[p * 3];
        pos[j * 3 + 1] = G[3] * o0 + G[4] * o1 + G[5] * o2 + pos[p * 3 + 1];
        pos[j * 3 + 2] = G[6] * o0 + G[7] * o1 + G[8] * o2 + pos[p * 3 + 2];
        double Rl[9];
        euler_to_rot(euler + 3 * j, Rl);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) Rg[j][r * 3 + c] = G[r * 3] * Rl[c] + G[r * 3 + 1] * Rl[3 + c] + G[r * 3 + 2] * Rl[6 + c];
    }
}

// skeleton joint <-> observation row (COCO-17 + mid-spine at 17); inverse_kinematics.py:366-378
MVMC_HD inline int ik_skel_of(int k) { const int t[16] = {1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15, 16, 17}; return t[k]; }
MVMC_HD inline int ik_obs_of(int k) { const int t[16] = {11, 13, 15, 12, 14, 16, 17, 5, 7, 9, 6, 8, 10, 0, 3, 4}; return t[k]; }

// residual functor of one IK stage: stage 0: x = [root, euler] (57) with the lengths fixed; stage 1: x = [root, euler, lengths]
struct IkResidual {
    const Skel* sk;
    const double* pose18;   // (nv,18,3): COCO-17 + mid-spine rows, x, y, score
    const double* Pm;       // (nv,3,4)
    const double* side_fixed;
    int nv, stage;
    MVMC_HD int m() const { return nv * 32; }
    MVMC_HD void operator()(const double* x, double* f) const {
        double pos[54];
        forward_kinematics(*sk, x, x + 3, stage == 0 ? side_fixed : x + 57, pos);
        for (int v = 0; v < nv; ++v) {
            const double* P = Pm + v * 12;
            for (int k = 0; k < 16; ++k) {
                const double* X = pos + ik_skel_of(k) * 3;
                const double h0 = P[0] * X[0] + P[1] * X[1] + P[2] * X[2] + P[3];
                const double h1 = P[4] * X[0] + P[5] * X[1] + P[6] * X[2] + P[7];
                const double h2 = P[8] * X[0] + P[9] * X[1] + P[10] * X[2] + P[11];
                const double w = 1e-5 + h2;
                const double* ob = pose18 + (v * 18 + ik_obs_of(k)) * 3;
                f[(v * 16 + k) * 2] = (h0 / w - ob[0]) * ob[2];
                f[(v * 16 + k) * 2 + 1] = (h1 / w - ob[1]) * ob[2];
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------------------------
// Cold start of PoseSolver.solve (inverse_kinematics.py:389-397): triangulate_point_groups_from_multiple_views_linear(projs, poses,
// 0.01, post_optimize=True) (mv_math_util.py:152-212) on the 18 observation rows, root = midpoint of the two hips.
// ---------------------------------------------------------------------------------------------------------------------------------

// DLT of observation row jo over the views (mv_math_util.py:165-187, :215-240): views with score >= min_score, all views when fewer
// than two qualify; the null vector of A^T A by cyclic Jacobi (A is 2V x 4).
MVMC_HD inline void dlt_point(const double* pose18, const double* Pm, int nv, int jo, double min_score, double* X) {
    int n_ok = 0;
    for (int v = 0; v < nv; ++v) n_ok += pose18[(v * 18 + jo) * 3 + 2] >= min_score;
    const bool use_all = n_ok < 2;
    double a[4][4], vv[4][4];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { a[r][c] = 0.0; vv[r][c] = (r == c) ? 1.0 : 0.0; }
    for (int v = 0; v < nv; ++v) {
        const double* kp = &pose18[(v * 18 + jo) * 3];
        if (!use_all && !(kp[2] >= min_score)) continue;
        const double* P = &Pm[v * 12];
        double r1[4], r2[4];
        for (int k = 0; k < 4; ++k) { r1[k] = kp[0] * P[8 + k] - P[k]; r2[k] = kp[1] * P[8 + k] - P[4 + k]; }
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) a[r][c] += r1[r] * r1[c] + r2[r] * r2[c];
    }
    const double tr = a[0][0] + a[1][1] + a[2][2] + a[3][3];
    for (int sweep = 0; sweep < 16; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 3; ++p) for (int q = p + 1; q < 4; ++q) off += a[p][q] * a[p][q];
        if (off <= 1e-36 * tr * tr) break;
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) {
                const double apq = a[p][q];
                if (fabs(apq) < 1e-300) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; ++k) { const double x = a[k][p], y = a[k][q]; a[k][p] = c * x - s * y; a[k][q] = s * x + c * y; }
                for (int k = 0; k < 4; ++k) { const double x = a[p][k], y = a[q][k]; a[p][k] = c * x - s * y; a[q][k] = s * x + c * y; }
                for (int k = 0; k < 4; ++k) { const double x = vv[k][p], y = vv[k][q]; vv[k][p] = c * x - s * y; vv[k][q] = s * x + c * y; }
            }
    }
    int m = 0;
    for (int k = 1; k < 4; ++k) if (a[k][k] < a[m][m]) m = k;
    X[0] = vv[0][m] / vv[3][m]; X[1] = vv[1][m] / vv[3][m]; X[2] = vv[2][m] / vv[3][m];
}

// residual of the post-optimisation (mv_math_util.py:191-201): x = n_pts points; f[v * n_pts + j] = |proj_v(X_j) - obs| * score
struct PostoptResidual {
    const double* pose;     // (nv, n_pts, 3)
    const double* Pm;       // (nv,3,4)
    int nv, n_pts;
    MVMC_HD int m() const { return nv * n_pts; }
    MVMC_HD void operator()(const double* x, double* f) const {
        for (int v = 0; v < nv; ++v) {
            const double* P = Pm + v * 12;
            for (int j = 0; j < n_pts; ++j) {
                const double* X = x + 3 * j;
                const double h0 = P[0] * X[0] + P[1] * X[1] + P[2] * X[2] + P[3];
                const double h1 = P[4] * X[0] + P[5] * X[1] + P[6] * X[2] + P[7];
                const double h2 = P[8] * X[0] + P[9] * X[1] + P[10] * X[2] + P[11];
                const double w = h2 + 1e-6;
                const double* ob = pose + (v * n_pts + j) * 3;
                const double du = h0 / w - ob[0], dv = h1 / w - ob[1];
                f[v * n_pts + j] = sqrt(du * du + dv * dv) * ob[2];
            }
        }
    }
};

// x (54) <- the 18 triangulated and post-optimised points; work as for trf() with (m, n) = (18 nv, 54)
template <class Ex>
MVMC_HD void triangulate_postopt18(const Ex& ex, const double* pose18, const double* Pm, int nv, double* x, double* work) {
    for (int j = ex.lane(); j < 18; j += ex.lanes()) dlt_point(pose18, Pm, nv, j, 0.01, x + 3 * j);
    ex.sync();
    PostoptResidual fun{pose18, Pm, nv, 18};
    trf(ex, fun, 54, fun.m(), x, 2, work);
}

}  // namespace trf_faithful
