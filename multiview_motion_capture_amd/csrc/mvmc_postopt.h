// Triangulation post-optimise: triangulate_point_groups_from_multiple_views_linear(post_optimize=True)
// (mv_math_util.py:189-210) = scipy least_squares(max_nfev = 2) on the unsigned residual
//     d[v][j] = |proj_v(X_j) - obs[v][j]| * score[v][j],      proj with eps = 1e-6,
// i.e. ONE trust-region trial step from the DLT points, kept only if it lowers the cost.
// The Jacobian is block diagonal (joint j only moves its own residuals), so J^T J is a set of 3x3
// blocks: one lane per joint diagonalises its block in registers and the trust-region root-find
// (solve_lsq_trust_region, common.py:57-168, both the full-rank and the rank-deficient branch) runs
// across the wave with shuffle reductions.  Analytic gradient instead of 2-point finite differences.
#pragma once
#include "mvmc_common.h"

namespace postopt {

constexpr double kEps = 2.220446049250313e-16;

template <int P, int Q>
__device__ __forceinline__ void rot3(double (&a)[3][3], double (&v)[3][3]) {
    const double apq = a[P][Q];
    if (fabs(apq) < 1e-300) return;
    const double theta = (a[Q][Q] - a[P][P]) / (2.0 * apq);
    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
    for (int k = 0; k < 3; ++k) { const double x = a[k][P], y = a[k][Q]; a[k][P] = c * x - s * y; a[k][Q] = s * x + c * y; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { const double x = a[P][k], y = a[Q][k]; a[P][k] = c * x - s * y; a[Q][k] = s * x + c * y; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { const double x = v[k][P], y = v[k][Q]; v[k][P] = c * x - s * y; v[k][Q] = s * x + c * y; }
}

// residuals of joint X against nv views; accumulates cost terms and (optionally) B = J^T J, g = J^T f
__device__ inline double joint_cost(const double* X, const double* obs /*[nv][stride]*/, int stride, const double* Pm,
                                    int nv, double (*B)[3], double* g) {
    double c2 = 0.0;
    for (int v = 0; v < nv; ++v) {
        const double* P = Pm + v * 12;
        const double* ob = obs + v * stride;
        const double h0 = P[0] * X[0] + P[1] * X[1] + P[2] * X[2] + P[3];
        const double h1 = P[4] * X[0] + P[5] * X[1] + P[6] * X[2] + P[7];
        const double w = P[8] * X[0] + P[9] * X[1] + P[10] * X[2] + P[11] + 1e-6;
        const double u = h0 / w, vv = h1 / w;
        const double ru = u - ob[0], rv = vv - ob[1], s = ob[2];
        const double rn = sqrt(ru * ru + rv * rv);
        const double f = rn * s;
        c2 += f * f;
        if (B) {
            double gr[3];
            for (int c = 0; c < 3; ++c) {
                const double du = (P[c] - u * P[8 + c]) / w, dv = (P[4 + c] - vv * P[8 + c]) / w;
                gr[c] = rn > 0.0 ? s * (ru * du + rv * dv) / rn : 0.0;
            }
            for (int a = 0; a < 3; ++a) {
                g[a] += gr[a] * f;
                for (int c = 0; c < 3; ++c) B[a][c] += gr[a] * gr[c];
            }
        }
    }
    return c2;
}

// One wave; lane j < J owns joint j (X in/out).  obs_j points at obs[0][j] (view stride `stride` doubles).
__device__ inline void post_optimize_wave(double* X, const double* obs_j, int stride, const double* Pm, int nv, int J) {
    const int lane = threadIdx.x & 63;
    const bool on = lane < J;
    double B[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, g[3] = {0, 0, 0}, Vm[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    double c2 = 0.0;
    if (on) c2 = joint_cost(X, obs_j, stride, Pm, nv, B, g);
    const double cost = 0.5 * wave_sum(c2);
    const double gmax = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
    double gm = on ? gmax : 0.0;
    for (int off = 32; off > 0; off >>= 1) gm = fmax(gm, __shfl_xor(gm, off, 64));
    if (gm < 1e-8) return;  // gtol
    double Delta = sqrt(wave_sum(on ? X[0] * X[0] + X[1] * X[1] + X[2] * X[2] : 0.0));
    if (Delta == 0.0) Delta = 1.0;
    // 3x3 eigen-decomposition of the joint's block
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = B[0][1] * B[0][1] + B[0][2] * B[0][2] + B[1][2] * B[1][2];
        const double tr = B[0][0] + B[1][1] + B[2][2];
        if (off <= 1e-36 * tr * tr) break;
        rot3<0, 1>(B, Vm); rot3<0, 2>(B, Vm); rot3<1, 2>(B, Vm);
    }
    double lam[3], suf[3];
    for (int k = 0; k < 3; ++k) {
        lam[k] = on ? fmax(B[k][k], 0.0) : 1.0;
        suf[k] = on ? Vm[0][k] * g[0] + Vm[1][k] * g[1] + Vm[2][k] * g[2] : 0.0;
    }
    // A joint that fewer than three views score has a rank-deficient block (one residual per view): its null direction is an exact
    // zero singular value with a zero projection in the reference's SVD of J.  In the normal-equation form the same direction shows
    // up as rounding noise (lam ~ 1e-16 lam_max with a noise gradient component), and because alpha ends near 0 whenever the
    // Gauss-Newton step is shorter than Delta, noise / (noise + alpha) would dominate the normalised step: drop it explicitly.
    {
        const double lmax = fmax(lam[0], fmax(lam[1], lam[2]));
        for (int k = 0; k < 3; ++k)
            if (lam[k] <= 1e-12 * lmax) { lam[k] = 0.0; suf[k] = 0.0; }
    }
    const int m = nv * J, n = 3 * J;
    double smax = 0.0, smin = 1e300;
    for (int k = 0; k < 3; ++k)
        if (on) { smax = fmax(smax, lam[k]); smin = fmin(smin, lam[k]); }
    for (int off = 32; off > 0; off >>= 1) { smax = fmax(smax, __shfl_xor(smax, off, 64)); smin = fmin(smin, __shfl_xor(smin, off, 64)); }
    const bool full_rank = (m >= n) && (sqrt(smin) > kEps * m * sqrt(smax));
    double coef[3];
    bool done = false;
    if (full_rank) {
        double pn = 0.0;
        for (int k = 0; k < 3; ++k) { coef[k] = on ? -suf[k] / lam[k] : 0.0; pn += coef[k] * coef[k]; }
        if (sqrt(wave_sum(pn)) <= Delta) done = true;  // Gauss-Newton step inside the region
    }
    if (!done) {
        const double sufn = sqrt(wave_sum(suf[0] * suf[0] + suf[1] * suf[1] + suf[2] * suf[2]));
        auto phi_of = [&](double alpha, double* phi_prime) {
            double a = 0.0, b = 0.0;
            for (int k = 0; k < 3; ++k) {
                const double den = lam[k] + alpha;
                if (suf[k] != 0.0) { a += (suf[k] / den) * (suf[k] / den); b += suf[k] * suf[k] / (den * den * den); }
            }
            const double pn = sqrt(wave_sum(a));
            *phi_prime = -wave_sum(b) / pn;
            return pn - Delta;
        };
        double alpha_upper = sufn / Delta;
        double alpha_lower = 0.0;
        if (full_rank) {
            double pp;
            const double phi = phi_of(0.0, &pp);
            alpha_lower = -phi / pp;
        }
        // initial_alpha = 0: kept as is when full rank (then reset inside the loop), seeded otherwise
        double alpha = full_rank ? 0.0 : fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
        for (int it = 0; it < 10; ++it) {
            if (alpha < alpha_lower || alpha > alpha_upper) alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
            double pp;
            const double phi = phi_of(alpha, &pp);
            if (phi < 0) alpha_upper = alpha;
            const double ratio = phi / pp;
            alpha_lower = fmax(alpha_lower, alpha - ratio);
            alpha -= (phi + Delta) * ratio / Delta;
            if (fabs(phi) < 0.01 * Delta) break;
        }
        // a last Newton update that overshoots below zero is clamped (mvmc_trf_faithful.h, solve_lsq_trust_region: SciPy's LAPACK
        // noise triplets stop it at ~ -1e-20): the step is then the minimum-norm Gauss-Newton step stretched to |p| = Delta.
        // Only where SciPy's thin SVD HAS such triplets: rank < min(m, n).  A two-view cluster whose joints are all scored by both
        // views (m = 2 J independent rows < n = 3 J) has none, nothing stops the last update, and the reference's step is the one with
        // the negative alpha of ordinary size that it leaves (found by tools/oracle_soak.py on a cluster of two false detections:
        // profiles/r05_oracle_soak.txt); with a joint that a view scores 0 (a zero row) the triplets are there and the clamp holds
        // (ik_cases case 7, tests/test_trf_faithful_cpu.py).
        int rk = 0;
        for (int k = 0; k < 3; ++k) rk += (on && lam[k] > 0.0) ? 1 : 0;
        const int rank = (int)wave_sum((double)rk);
        if (!full_rank && alpha < 0.0 && rank < (m < n ? m : n)) alpha = 0.0;
        double pn = 0.0;
        for (int k = 0; k < 3; ++k) {
            coef[k] = (on && suf[k] != 0.0) ? -suf[k] / (lam[k] + alpha) : 0.0;
            pn += coef[k] * coef[k];
        }
        const double sc = Delta / sqrt(wave_sum(pn));
        for (int k = 0; k < 3; ++k) coef[k] *= sc;
    }
    double Xn[3];
    for (int r = 0; r < 3; ++r) Xn[r] = X[r] + Vm[r][0] * coef[0] + Vm[r][1] * coef[1] + Vm[r][2] * coef[2];
    const double c2n = on ? joint_cost(Xn, obs_j, stride, Pm, nv, nullptr, nullptr) : 0.0;
    const double cost_new = 0.5 * wave_sum(c2n);
    if (isfinite(cost_new) && cost - cost_new > 0.0 && on) { X[0] = Xn[0]; X[1] = Xn[1]; X[2] = Xn[2]; }
}

}  // namespace postopt
