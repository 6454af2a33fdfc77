// Temporal inverse kinematics for gfx950: PoseSolver.solve (inverse_kinematics.py:351-433).
//
// One 256-thread workgroup (4 waves) solves one person-frame; the whole solver state lives in LDS.
// Two trust-region stages, each a restatement of SciPy's trf_no_bounds / solve_lsq_trust_region as the
// reference reaches them through least_squares(fun, x0, max_nfev=k) (trf.py:401-560, common.py:57-168):
//   stage 1  x = [root, euler(18x3)]                (57 parameters, 39 structurally non-null)
//   stage 2  x = [root, euler, side bone lengths]   (68 parameters, 49 structurally non-null)
// Differences from the reference that are deliberate (DESIGN.md "IK parity"):
//   * analytic Jacobian of the FK chain instead of 2-point finite differences;
//   * the trust-region sub-problem is solved from the normal equations J^T J = D^T W D (D = d pos / d x is
//     48 x n, W the per-joint 3x3 image-space blocks).  J^T J is assembled straight into registers and
//     tridiagonalised from g = J^T f (Householder form of the Lanczos process, mvmc_eigh_tri.h): the Krylov space
//     of (J^T J, g) contains every p(alpha) = -(J^T J + alpha)^-1 g and none of the Jacobian's null directions
//     (bone twists etc., 9 of the 39 / 49 columns), so SciPy's Newton iteration on |p(alpha)| = Delta runs on a
//     small positive-definite tridiagonal matrix -- no eigendecomposition.  When the split between range and
//     null space is not clean (weakly observed directions, missing joints) the step falls back to the full
//     eigensolver, where p(alpha) = -V (V^T g / (L + alpha)) with the null cluster removed;
//   * columns that are identically zero for every input (leaf-joint angles, the root's bone length) are
//     removed from the problem -- SciPy gives them s = 0 and a zero step, which is what they get here.
#include <stdlib.h>

#include "mvmc_common.h"
#include "mvmc_postopt.h"
#include "mvmc_eigh_tri.h"
#include "mvmc_ik_shared.h"

namespace {

#ifdef MVMC_IK_PROFILE
#define PROF_T0 const long long _t0 = clock64();
#define PROF_ADD(S, k) if (threadIdx.x == 0) (S).prof[k] += clock64() - _t0;
#else
#define PROF_T0
#define PROF_ADD(S, k)
#endif

constexpr int NT = 256;    // threads per problem
constexpr int NA = 50;     // max active parameters (even)
constexpr int LD = 51;     // odd leading dimension: conflict-free column walks on 8-byte elements
constexpr int LDZ = LD;    // eigenvector rows of the fallback eigensolver (odd: lanes walk one column)
constexpr int NROW = 48;   // rows of D: 16 observed joints x 3
// bufB = ik_eval's per-(view,joint) Jacobian blocks [0, EVS) + solver vectors (64 doubles each; SV and PART longer)
constexpr int EVS = 8 * 16 * 10;
constexpr int SCR0 = EVS;
enum { SC_DSC = 0, SC_E2, SC_LMUL, SC_DINV, SC_RH, SC_TAU, SC_D, SC_E, SC_QC, SC_V0, SC_WN, SC_SV /* 2 slots */, SC_SV2,
       SC_PART /* 4 slots */, SC_PART2, SC_PART3, SC_PART4, SC_COUNT };
constexpr int BUFA = 48 * eightri::KLD;          // D (48 x LD) during the model build, Householder vectors after
constexpr int BUFB = EVS + SC_COUNT * 64;
static_assert(BUFA >= 48 * LD, "D does not fit bufA");
// fallback eigensolver: three n x n matrices per workgroup in global memory (include/mvmc.h: MVMC_IK_SCRATCH_DOUBLES)
constexpr int FBM = 2560;
static_assert(FBM >= NA * LD && 3 * FBM == MVMC_IK_SCRATCH_DOUBLES, "fallback scratch size");

struct IkShared {
    double bufA[BUFA];     // D (48 x LD) | Householder vectors of the tridiagonalisation (row k = v_k, KLD apart)
    double bufB[BUFB];     // per-(view,joint) Jacobian blocks [0, EVS) | solver vectors [SCR0, ...)
    double x[68], xn[68];
    double side[18];       // side bone lengths used by stage 1 (fixed)
    double g[NA], cv[NA], step[NA];
    double obs[VMAX * NOBS * 3], Pm[VMAX * 12];
    double Rl[18 * 9], Rg[18 * 9], pos[18 * 3], bvec[18 * 3], off[18 * 3], axes[18 * 9];
    double Wk[NOBS * 6], tk[NOBS * 3];
    double red[8];
    double sc[8];          // broadcast scalars
    // skeleton tables (copied from the kernel argument once: dynamic indexing of by-value kernel
    // arguments costs SGPR spills and scratch)
    double dirs[18 * 3], ref_side[18];
    int anc[18], depth[18], maxdepth, nviews, na[2], n_side;
    int parents[18], side_map[18];
    unsigned char act[2][NA], colkind[2][NA], cola[2][NA], colc[2][NA];
    signed char inv_act[2][68];
#ifdef MVMC_IK_PROFILE
    long long prof[8];
#endif
};

struct SkelRef {  // what the solver routines see
    const double (*dirs)[3];
    const int* parents;
    const int* side_map;
    int n_side;
};

__device__ inline double block_sum256(double v, double* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------------------------------------
// FK + residual (+ per-joint normal-equation blocks when want_jac).  Returns the cost 0.5*|f|^2.
// ---------------------------------------------------------------------------------------------
__device__ __noinline__ double ik_eval(IkShared& S, const SkelRef& sk, const double* xs, int stage, bool want_jac) {
    const int tid = threadIdx.x;
    if (tid < 18) {
        euler_to_rot(xs + 3 + 3 * tid, &S.Rl[tid * 9]);
        double len = 0.0;
        if (tid > 0) len = (stage == 0) ? S.side[sk.side_map[tid]] : xs[57 + sk.side_map[tid]];
        for (int k = 0; k < 3; ++k) S.off[tid * 3 + k] = sk.dirs[tid][k] * len;
    }
    __syncthreads();
    if (tid < 9) S.Rg[tid] = S.Rl[tid];
    if (tid < 3) S.pos[tid] = xs[tid];
    __syncthreads();
    for (int lev = 1; lev <= S.maxdepth; ++lev) {
        if (tid < 162) {
            const int j = tid / 9, e = tid - j * 9;
            if (S.depth[j] == lev) {
                const int p = sk.parents[j], r = e / 3, c = e - r * 3;
                const double* Gp = &S.Rg[p * 9];
                const double* Rj = &S.Rl[j * 9];
                S.Rg[j * 9 + e] = Gp[r * 3] * Rj[c] + Gp[r * 3 + 1] * Rj[3 + c] + Gp[r * 3 + 2] * Rj[6 + c];
                if (e < 3) {
                    S.pos[j * 3 + e] = Gp[e * 3] * S.off[j * 3] + Gp[e * 3 + 1] * S.off[j * 3 + 1] +
                                       Gp[e * 3 + 2] * S.off[j * 3 + 2] + S.pos[p * 3 + e];
                    S.bvec[j * 3 + e] = Gp[e * 3] * sk.dirs[j][0] + Gp[e * 3 + 1] * sk.dirs[j][1] +
                                        Gp[e * 3 + 2] * sk.dirs[j][2];
                }
            }
        }
        __syncthreads();
    }
    // residuals: one thread per (view, observed joint)
    double f2 = 0.0;
    const int nvk = S.nviews * NOBS;
    if (tid < nvk) {
        const int v = tid / NOBS, k = tid - v * NOBS;
        const double* X = &S.pos[kIkSkel[k] * 3];
        const double* P = &S.Pm[v * 12];
        const double h0 = P[0] * X[0] + P[1] * X[1] + P[2] * X[2] + P[3];
        const double h1 = P[4] * X[0] + P[5] * X[1] + P[6] * X[2] + P[7];
        const double h2 = P[8] * X[0] + P[9] * X[1] + P[10] * X[2] + P[11];
        const double w = 1e-5 + h2, iw = 1.0 / w;
        const double u = h0 / w, vv = h1 / w;
        const double* ob = &S.obs[tid * 3];
        const double s = ob[2];
        const double fu = (u - ob[0]) * s, fv = (vv - ob[1]) * s;
        f2 = fu * fu + fv * fv;
        if (want_jac) {
            double du[3], dv[3];
            for (int c = 0; c < 3; ++c) {
                du[c] = (P[c] - u * P[8 + c]) * iw;
                dv[c] = (P[4 + c] - vv * P[8 + c]) * iw;
            }
            const double s2 = s * s;
            double* o = &S.bufB[tid * 10];
            o[0] = s2 * (du[0] * du[0] + dv[0] * dv[0]);
            o[1] = s2 * (du[0] * du[1] + dv[0] * dv[1]);
            o[2] = s2 * (du[0] * du[2] + dv[0] * dv[2]);
            o[3] = s2 * (du[1] * du[1] + dv[1] * dv[1]);
            o[4] = s2 * (du[1] * du[2] + dv[1] * dv[2]);
            o[5] = s2 * (du[2] * du[2] + dv[2] * dv[2]);
            o[6] = s * (du[0] * fu + dv[0] * fv);
            o[7] = s * (du[1] * fu + dv[1] * fv);
            o[8] = s * (du[2] * fu + dv[2] * fv);
        }
    }
    return 0.5 * block_sum256(f2, S.red);
}

// ---------------------------------------------------------------------------------------------
// Jacobian model at the accepted point.  Needs ik_eval(want_jac) state (blocks in bufB, FK arrays).
//   ik_model:        W_k, t_k, rotation axes, D -> bufA (48 x nap, zero padded), g = D^T t
//   ik_normal_matrix bufB <- D^T W D (fallback eigensolver path; bufA must hold D)
// ---------------------------------------------------------------------------------------------
__device__ void ik_build_D(IkShared& S, const SkelRef& sk, int stage) {
    const int tid = threadIdx.x;
    const int na = S.na[stage], nap = (na + 1) & ~1;
    // D[row = 3k + c3][col] = d pos_K[c3] / d x_col
    for (int idx = tid; idx < NROW * nap; idx += NT) {
        const int row = idx / nap, col = idx - row * nap;
        const int k = row / 3, c3 = row - k * 3, K = kIkSkel[k];
        double d = 0.0;
        if (col < na) {
            const int kind = S.colkind[stage][col], a = S.cola[stage][col], c = S.colc[stage][col];
            if (kind == 0) {
                d = (c3 == c) ? 1.0 : 0.0;
            } else if (kind == 1) {
                if ((S.anc[K] >> a) & 1) {
                    const double* ax = &S.axes[(a * 3 + c) * 3];
                    const double r0 = S.pos[K * 3] - S.pos[a * 3], r1 = S.pos[K * 3 + 1] - S.pos[a * 3 + 1],
                                 r2 = S.pos[K * 3 + 2] - S.pos[a * 3 + 2];
                    d = (c3 == 0) ? ax[1] * r2 - ax[2] * r1 : (c3 == 1) ? ax[2] * r0 - ax[0] * r2 : ax[0] * r1 - ax[1] * r0;
                }
            } else {
                for (int j = K; j > 0; j = sk.parents[j])
                    if (sk.side_map[j] == a) d += S.bvec[j * 3 + c3];
            }
        }
        S.bufA[row * LD + col] = d;
    }
    __syncthreads();
}

// rotation axes in the world frame: R_a = Rx Ry Rz inside the parent's frame.  Out of line (like ik_eval): the
// double-precision sincos expansions carry a dozen 64-bit constants that the compiler otherwise keeps live --
// and spills -- across the whole kernel.
__device__ __noinline__ void ik_rotation_axes(IkShared& S, const SkelRef& sk, const double* xs) {
    const int tid = threadIdx.x;
    if (tid >= 192 && tid < 192 + 54) {
        const int t = tid - 192, a = t / 3, c = t - a * 3;
        const double* e = xs + 3 + 3 * a;
        double l[3];
        if (c == 0) { l[0] = 1; l[1] = 0; l[2] = 0; }
        else if (c == 1) { double s0, c0; sincos(e[0], &s0, &c0); l[0] = 0; l[1] = c0; l[2] = s0; }
        else {
            double s0, c0, s1, c1;
            sincos(e[0], &s0, &c0); sincos(e[1], &s1, &c1);
            l[0] = s1; l[1] = -s0 * c1; l[2] = c0 * c1;
        }
        if (a == 0) {
            for (int r = 0; r < 3; ++r) S.axes[t * 3 + r] = l[r];
        } else {
            const double* Gp = &S.Rg[sk.parents[a] * 9];
            for (int r = 0; r < 3; ++r) S.axes[t * 3 + r] = Gp[r * 3] * l[0] + Gp[r * 3 + 1] * l[1] + Gp[r * 3 + 2] * l[2];
        }
    }
}

__device__ void ik_model(IkShared& S, const SkelRef& sk, const double* xs, int stage) {
    const int tid = threadIdx.x;
    const int na = S.na[stage], nap = (na + 1) & ~1;
    // per-joint blocks: W_k = sum_v s^2 (du du^T + dv dv^T), t_k = sum_v s (du fu + dv fv)
    if (tid < NOBS * 9) {
        const int k = tid / 9, e = tid - k * 9;
        double a = 0.0;
        for (int v = 0; v < S.nviews; ++v) a += S.bufB[(v * NOBS + k) * 10 + e];
        if (e < 6) S.Wk[k * 6 + e] = a; else S.tk[k * 3 + (e - 6)] = a;
    }
    ik_rotation_axes(S, sk, xs);
    __syncthreads();
    ik_build_D(S, sk, stage);
    if (tid < nap) {
        double a = 0.0;
        for (int row = 0; row < NROW; ++row) a += S.bufA[row * LD + tid] * S.tk[row];
        S.g[tid] = a;
    }
    __syncthreads();
}

// Fallback only: Y = W D, then J^T J = D^T Y, both in the workgroup's global scratch (bufA must hold D)
__device__ void ik_normal_matrix(IkShared& S, int stage, double* Y, double* A) {
    const int tid = threadIdx.x;
    const int na = S.na[stage], nap = (na + 1) & ~1;
    for (int idx = tid; idx < NROW * nap; idx += NT) {
        const int row = idx / nap, col = idx - row * nap;
        const int k = row / 3, c3 = row - k * 3;
        const double* W = &S.Wk[k * 6];
        const double w0 = (c3 == 0) ? W[0] : (c3 == 1) ? W[1] : W[2];
        const double w1 = (c3 == 0) ? W[1] : (c3 == 1) ? W[3] : W[4];
        const double w2 = (c3 == 0) ? W[2] : (c3 == 1) ? W[4] : W[5];
        Y[row * LD + col] = w0 * S.bufA[(3 * k) * LD + col] + w1 * S.bufA[(3 * k + 1) * LD + col] +
                            w2 * S.bufA[(3 * k + 2) * LD + col];
    }
    __syncthreads();
    for (int idx = tid; idx < nap * nap; idx += NT) {
        const int i = idx / nap, j = idx - i * nap;
        double a = 0.0;
        for (int row = 0; row < NROW; ++row) a += S.bufA[row * LD + i] * Y[row * LD + j];
        A[i * LD + j] = a;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Fast path: J^T J = D^T (W D) into registers, Krylov tridiagonalisation from g, leading-block checks.
// Returns the size of the leading (range) block, or -1 (uniformly) when the eigensolver has to take over.
// S.sc[4..7] = {beta0, tau0, |J^T J|_inf, coupling}.
// ---------------------------------------------------------------------------------------------
__device__ __noinline__ int ik_krylov_model(IkShared& S, int stage) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int na = S.na[stage];
    double* scr = S.bufB + SCR0;
#ifdef MVMC_IK_PROFILE
    long long _tp = clock64();
#define KSTAMP(k) { const long long _t = clock64(); if (tid == 0) S.prof[k] += _t - _tp; _tp = _t; }
#else
#define KSTAMP(k)
#endif
    // a[q] = (J^T J)[w + 4 q][lane] = sum over joints k of D_k[:, i]^T (W_k D_k[:, lane]), the weighted column
    // formed on the fly
    double a[eightri::KQ];
    int ic[eightri::KQ];
#pragma unroll
    for (int q = 0; q < eightri::KQ; ++q) {
        const int i = w + 4 * q;
        ic[q] = i < na ? i : na - 1;
        a[q] = 0.0;
    }
    const int jc = lane < na ? lane : na - 1;
    for (int k = 0; k < NOBS; ++k) {
        const double* W = &S.Wk[k * 6];
        const double* D0 = &S.bufA[(3 * k) * LD];
        const double d0 = D0[jc], d1 = D0[LD + jc], d2 = D0[2 * LD + jc];
        const double y0 = W[0] * d0 + W[1] * d1 + W[2] * d2;
        const double y1 = W[1] * d0 + W[3] * d1 + W[4] * d2;
        const double y2 = W[2] * d0 + W[4] * d1 + W[5] * d2;
#pragma unroll
        for (int q = 0; q < eightri::KQ; ++q)
            a[q] += D0[ic[q]] * y0 + D0[LD + ic[q]] * y1 + D0[2 * LD + ic[q]] * y2;
    }
#pragma unroll
    for (int q = 0; q < eightri::KQ; ++q)
        if (w + 4 * q >= na || lane >= na) a[q] = 0.0;
    __syncthreads();  // D is dead from here: bufA takes the Householder vectors
    KSTAMP(4)
    const int kk = eightri::tridiag_krylov(a, S.g, S.bufA, na, scr + 64 * SC_D, scr + 64 * SC_E, scr + 64 * SC_TAU,
                                           scr + 64 * SC_V0, scr + 64 * SC_SV, scr + 64 * SC_PART, S.red, &S.sc[4]);
    KSTAMP(5)
    if (tid < 64) {
        const bool ok = kk > 0 && eightri::krylov_block_ok(scr + 64 * SC_D, scr + 64 * SC_E, kk, na, S.sc[6], S.sc[7],
                                                           scr + 64 * SC_DSC, scr + 64 * SC_E2, scr + 64 * SC_LMUL,
                                                           scr + 64 * SC_DINV, scr + 64 * SC_WN);
        scr[64 * SC_RH + tid] = tid == 0 ? S.sc[4] : 0.0;
        if (tid == 0) S.sc[0] = ok ? (double)kk : -1.0;
    }
    __syncthreads();
    KSTAMP(6)
    return (int)S.sc[0];
}

// ---------------------------------------------------------------------------------------------
// Fallback model: no clean split between range and null space -> eigendecomposition of J^T J with the null
// cluster removed (lam, eigenvector rows in bufC, suf = V^T g).  Kept out of line: it runs for a fraction of a
// percent of the models and would otherwise set the register budget of the whole kernel.
// ---------------------------------------------------------------------------------------------
__device__ __noinline__ void ik_eigen_model(IkShared& S, const SkelRef& sk, int stage, double* fb) {
    const int tid = threadIdx.x;
    const int na = S.na[stage], nap = (na + 1) & ~1;
    double* scr = S.bufB + SCR0;
    double *gA = fb, *gZ = fb + FBM, *gW = fb + 2 * FBM;
    ik_build_D(S, sk, stage);               // bufA held Householder vectors
    ik_normal_matrix(S, stage, gW, gA);
    // lam -> SC_D, suf -> SC_E (they live until the next model); the other slots are eigensolver temporaries
    eightri::eigh(gA, LD, gZ, LDZ, gW, LD, nap, scr + 64 * SC_D, scr + 64 * SC_TAU, scr + 64 * SC_RH, scr + 64 * SC_V0,
                  scr + 64 * SC_WN, scr + 64 * SC_SV, S.red, reinterpret_cast<int*>(scr + 64 * SC_PART));
    if (tid < nap) {
        double a = 0.0;
        for (int i = 0; i < nap; ++i) a += gZ[tid * LDZ + i] * S.g[i];
        scr[64 * SC_E + tid] = a;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// solve_lsq_trust_region (common.py:57-168) in the eigenbasis of J^T J (fallback path), one wave.
// Every problem is rank deficient in the reference's terms (it keeps the structurally-zero columns, s = 0),
// so only that branch exists: Newton on phi(alpha) = |p(alpha)| - Delta from alpha0, the step normalised to
// |p| = Delta.  S.cv <- coefficients; returns alpha.
// ---------------------------------------------------------------------------------------------
__device__ double ik_tr_solve(IkShared& S, int nap, double Delta, double alpha0, double gg, double* pred, double* pnorm) {
    const int lane = threadIdx.x;
    const bool on = lane <= nap;
    const double* scr = S.bufB + SCR0;
    double lam = lane < nap ? scr[64 * SC_D + lane] : 1.0, suf = lane < nap ? scr[64 * SC_E + lane] : 0.0;
    // Virtual absorber (lane nap): a direction with lambda = 0 and a small fixed weight.  In the reference
    // the numerically-null directions of J carry finite-difference noise (s*u^T f ~ 1e-8 |g|), and because
    // SciPy normalises every rank-deficient step to |p| = Delta, that noise soaks up whatever part of the
    // trust-region length the Gauss-Newton step does not use -- in directions that do not move the skeleton.
    // The absorber plays that role deterministically and its coefficient is dropped from the step.
    if (lane == nap) {
        lam = 0.0;
        suf = 1e-8 * sqrt(gg);
    }
    double alpha_upper = sqrt(wave_sum(suf * suf)) / Delta;
    double alpha_lower = 0.0;
    double alpha = (alpha0 == 0.0) ? fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper)) : alpha0;
    for (int it = 0; it < 10; ++it) {
        if (alpha < alpha_lower || alpha > alpha_upper)
            alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
        const double denom = lam + alpha;
        const double t = suf != 0.0 ? suf / denom : 0.0;
        const double p_norm = sqrt(wave_sum(t * t));
        const double phi = p_norm - Delta;
        const double phi_prime = -wave_sum(suf != 0.0 ? suf * suf / (denom * denom * denom) : 0.0) / p_norm;
        if (phi < 0) alpha_upper = alpha;
        const double ratio = phi / phi_prime;
        alpha_lower = fmax(alpha_lower, alpha - ratio);
        alpha -= (phi + Delta) * ratio / Delta;
        if (fabs(phi) < 0.01 * Delta) break;
    }
    double c = (on && suf != 0.0) ? -suf / (lam + alpha) : 0.0;
    const double pn = sqrt(wave_sum(c * c));
    c *= Delta / pn;
    if (lane < nap) S.cv[lane] = c;
    *pred = -(0.5 * wave_sum(lam * c * c) + wave_sum(suf * c));
    *pnorm = sqrt(wave_sum(c * c));  // |step_h| including the absorber (== Delta up to rounding, as in SciPy)
    return alpha;
}

// ---------------------------------------------------------------------------------------------
// The same routine on the leading block of the Krylov tridiagonal matrix (fast path), one wave: Newton on
// phi(alpha) with cyclic-reduction solves, then the step back in parameter space (scratch slot SC_QC).
// Out of line: its register needs (PCR multipliers of six rounds) would otherwise set the budget of the loop
// around it.
// ---------------------------------------------------------------------------------------------
__device__ __noinline__ double ik_tr_solve_krylov(IkShared& S, int kk, int na, double Delta, double alpha0, double gg,
                                                  double pivmin, double tau0, double coupling, double* pred, double* pnorm) {
    const int tid = threadIdx.x;
    double* scr = S.bufB + SCR0;
    const double al = eightri::tr_solve_tri<false>(scr + 64 * SC_D, scr + 64 * SC_E, scr + 64 * SC_RH, kk, Delta, alpha0, gg,
                                                   pivmin, nullptr, nullptr, nullptr, nullptr, S.cv, pred, pnorm);
    double c = tid < kk ? S.cv[tid] : 0.0;
    if (kk < na) {
        // component along the first null coordinate: keeps the step orthogonal to the null vector
        const double eta = coupling * wave_sum_dpp(tid < kk ? scr[64 * SC_WN + tid] * c : 0.0);
        if (tid == kk) c = eta;
    }
    scr[64 * SC_QC + tid] = eightri::apply_q_krylov(S.bufA, scr + 64 * SC_TAU, scr + 64 * SC_V0, tau0, kk, na, c);
    return al;
}

// ---------------------------------------------------------------------------------------------
// trf_no_bounds (trf.py:401-560) with x_scale = 1, linear loss, ftol = xtol = gtol = 1e-8.
// ---------------------------------------------------------------------------------------------
__device__ void ik_trf(IkShared& S, const SkelRef& sk, int stage, int max_nfev, double* fb, double* cost_out, int* nfev_out,
                       int* njev_out, int* status_out, int* fallbacks_out) {
    const int tid = threadIdx.x;
    const int nfull = (stage == 0) ? 57 : 57 + sk.n_side;
    const int na = S.na[stage], nap = (na + 1) & ~1;
    const double ftol = 1e-8, xtol = 1e-8, gtol = 1e-8;
    double* scr = S.bufB + SCR0;

    double cost;
    { PROF_T0 cost = ik_eval(S, sk, S.x, stage, true); PROF_ADD(S, 0) }
    { PROF_T0 ik_model(S, sk, S.x, stage); PROF_ADD(S, 1) }
    int nfev = 1, njev = 1, status = -1;
    double xx = (tid < nfull) ? S.x[tid] * S.x[tid] : 0.0;
    double Delta = sqrt(block_sum256(xx, S.red));
    if (Delta == 0.0) Delta = 1.0;
    double alpha = 0.0;

    while (true) {
        // |g|_inf over the active set (null columns have g = 0)
        if (tid == 0) {
            double gm = 0.0;
            for (int i = 0; i < na; ++i) gm = fmax(gm, fabs(S.g[i]));
            S.sc[0] = gm;
        }
        __syncthreads();
        if (S.sc[0] < gtol) status = 1;
        if (status != -1 || nfev == max_nfev) break;

        const double gg = block_sum256(tid < nap ? S.g[tid] * S.g[tid] : 0.0, S.red);
        int kk;
        {
            PROF_T0
            kk = ik_krylov_model(S, stage);
            if (kk < 0) {
                ik_eigen_model(S, sk, stage, fb);
                ++*fallbacks_out;
            }
            PROF_ADD(S, 2)
        }
        const bool fast = kk > 0;
        const double pivmin = 1e-16 * S.sc[6] + 1e-300, tau0 = S.sc[5], coupling = S.sc[7];

        double actual = -1.0, cost_new = cost;
        while (actual <= 0.0 && nfev < max_nfev) {
            {
                PROF_T0
                if (tid < 64) {
                    double pred, pnorm, al;
                    if (fast) {
                        al = ik_tr_solve_krylov(S, kk, na, Delta, alpha, gg, pivmin, tau0, coupling, &pred, &pnorm);
                    } else {
                        al = ik_tr_solve(S, nap, Delta, alpha, gg, &pred, &pnorm);
                    }
                    if (tid == 0) { S.sc[1] = al; S.sc[2] = pred; S.sc[3] = pnorm; }
                }
                __syncthreads();
                if (tid < nap) {
                    double a = 0.0;
                    if (!fast) {
                        for (int j = 0; j < nap; ++j) a += fb[FBM + j * LDZ + tid] * S.cv[j];
                    } else {
                        a = scr[64 * SC_QC + tid];
                    }
                    S.step[tid] = a;
                }
                __syncthreads();
                PROF_ADD(S, 3)
            }
            alpha = S.sc[1];
            const double pred = S.sc[2];
            const double step_norm = S.sc[3];
            xx = (tid < nfull) ? S.x[tid] * S.x[tid] : 0.0;
            const double x_norm = sqrt(block_sum256(xx, S.red));
            if (tid < nfull) {
                const int ia = S.inv_act[stage][tid];
                S.xn[tid] = S.x[tid] + (ia >= 0 ? S.step[ia] : 0.0);
            }
            __syncthreads();
            { PROF_T0 cost_new = ik_eval(S, sk, S.xn, stage, true); PROF_ADD(S, 0) }
            ++nfev;
            if (!isfinite(cost_new)) { Delta = 0.25 * step_norm; continue; }
            actual = cost - cost_new;
            // update_tr_radius (common.py:222-245)
            double ratio;
            if (pred > 0.0) ratio = actual / pred;
            else if (pred == 0.0 && actual == 0.0) ratio = 1.0;
            else ratio = 0.0;
            double Delta_new = Delta;
            if (ratio < 0.25) Delta_new = 0.25 * step_norm;
            else if (ratio > 0.75 && step_norm > 0.95 * Delta) Delta_new = Delta * 2.0;
            // check_termination (common.py:705-717)
            const bool f_ok = (actual < ftol * cost) && (ratio > 0.25);
            const bool x_ok = step_norm < xtol * (xtol + x_norm);
            if (f_ok && x_ok) status = 4; else if (f_ok) status = 2; else if (x_ok) status = 3;
            if (status != -1) break;
            alpha *= Delta / Delta_new;
            Delta = Delta_new;
        }
        if (actual > 0.0) {
            __syncthreads();
            if (tid < nfull) S.x[tid] = S.xn[tid];
            __syncthreads();
            cost = cost_new;
            if (status == -1 && nfev < max_nfev) {
                // the accepted point's FK / residual blocks are still in LDS (last ik_eval was at xn)
                { PROF_T0 ik_model(S, sk, S.x, stage); PROF_ADD(S, 1) }
                ++njev;
            }
        }
    }
    if (status == -1) status = 0;
    *cost_out = cost; *nfev_out = nfev; *njev_out = njev; *status_out = status;
}

// Cold start: DLT of the 18 keypoints + the reference's one-step post-optimisation, hips -> S.xn[0..6).  Out of line
// (chain heads only; its 4x4 Jacobi arrays should not weigh on the solver's register budget).
__device__ __noinline__ void ik_cold_root(IkShared& S, const double* pose18, int nv) {
    const int tid = threadIdx.x;
    if (tid < 64) {
        double X[3] = {0, 0, 0};
        if (tid < 18) dlt_obs_point(pose18, S.Pm, nv, tid, 0.01, X);
        postopt::post_optimize_wave(X, pose18 + (tid < 18 ? tid : 0) * 3, 54, S.Pm, nv, 18);
        if (tid == 11 || tid == 12)
            for (int c = 0; c < 3; ++c) S.xn[(tid - 11) * 3 + c] = X[c];
    }
}

__global__ void __launch_bounds__(NT, 3)
ik_kernel(SkelDev skarg, const double* __restrict__ kps17, const double* __restrict__ Pmats,
          const int32_t* __restrict__ members, int B, int V, int C, int Pmax, const double* __restrict__ init,
          const uint8_t* __restrict__ cold, int nfev_cold, int nfev_warm, double* __restrict__ params_out,
          double* __restrict__ joints_out, double* __restrict__ info_out, double* __restrict__ scratch) {
    __shared__ IkShared S;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n_side = skarg.n_side;
    if (tid < 18) {
        for (int k = 0; k < 3; ++k) S.dirs[tid * 3 + k] = skarg.dirs[tid][k];
        S.parents[tid] = skarg.parents[tid];
        S.side_map[tid] = skarg.side_map[tid];
        S.ref_side[tid] = skarg.ref_side[tid];
    }
    __syncthreads();
    SkelRef sk;
    sk.dirs = reinterpret_cast<const double(*)[3]>(S.dirs);
    sk.parents = S.parents;
    sk.side_map = S.side_map;
    sk.n_side = n_side;

    // ---- static tables: depth, ancestor masks, active columns of both stages ----
    if (tid == 0) {
        int md = 0;
        for (int j = 0; j < 18; ++j) {
            int d = 0, m = 0;
            for (int a = sk.parents[j]; a >= 0; a = sk.parents[a]) { ++d; m |= 1 << a; }
            S.depth[j] = d; S.anc[j] = m;
            if (d > md) md = d;
        }
        S.maxdepth = md;
        int moved = 0, lens = 0;  // joints whose rotation moves an observed joint; used length slots
        for (int k = 0; k < NOBS; ++k) {
            const int K = kIkSkel[k];
            moved |= S.anc[K];
            for (int j = K; j > 0; j = sk.parents[j]) {
                const double* d = sk.dirs[j];
                if (d[0] != 0.0 || d[1] != 0.0 || d[2] != 0.0) lens |= 1 << sk.side_map[j];
            }
        }
        for (int st = 0; st < 2; ++st) {
            int n = 0;
            for (int i = 0; i < 68; ++i) S.inv_act[st][i] = -1;
            for (int c = 0; c < 3; ++c) { S.act[st][n] = c; S.colkind[st][n] = 0; S.cola[st][n] = 0; S.colc[st][n] = c; ++n; }
            for (int a = 0; a < 18; ++a)
                if ((moved >> a) & 1)
                    for (int c = 0; c < 3 && n < NA; ++c) {
                        S.act[st][n] = 3 + 3 * a + c; S.colkind[st][n] = 1; S.cola[st][n] = a; S.colc[st][n] = c; ++n;
                    }
            if (st == 1)
                for (int s = 0; s < n_side && n < NA; ++s)
                    if ((lens >> s) & 1) { S.act[st][n] = 57 + s; S.colkind[st][n] = 2; S.cola[st][n] = s; S.colc[st][n] = 0; ++n; }
            S.na[st] = n;
            for (int i = 0; i < n; ++i) S.inv_act[st][S.act[st][i]] = i;
        }
        int nv = 0;
        for (int v = 0; v < V && nv < VMAX; ++v) nv += members[(size_t)b * V + v] >= 0;
        S.nviews = nv;
    }
    __syncthreads();
    const int nv = S.nviews;
    double* info = info_out ? info_out + (size_t)b * 8 : nullptr;
    if (nv < 2) {  // the reference only solves clusters with >= 2 views (motion_capture.py:927,940)
        const double nan = __longlong_as_double(0x7ff8000000000000LL);
        for (int i = tid; i < 68; i += NT) params_out[(size_t)b * 68 + i] = nan;
        for (int i = tid; i < 54; i += NT) joints_out[(size_t)b * 54 + i] = nan;
        if (info && tid < 8) info[tid] = nan;
        return;
    }

    // ---- gather observations: 17 COCO rows + synthetic mid-spine (inverse_kinematics.py:339-348) ----
    double* pose18 = S.bufA;  // [nv][18][3] scratch
    if (tid < nv) {
        int seen = 0, q = -1;
        for (int v = 0; v < V; ++v) {
            const int m = members[(size_t)b * V + v];
            if (m >= 0) { if (seen == tid) { q = m; break; } ++seen; }
        }
        const double* kp = kps17 + (size_t)q * 51;
        double* dst = pose18 + tid * 54;
        for (int e = 0; e < 51; ++e) dst[e] = kp[e];
        // COCO: L_Shoulder 5, R_Shoulder 6, L_Hip 11, R_Hip 12
        for (int c = 0; c < 2; ++c) {
            const double mid_sh = 0.5 * (kp[5 * 3 + c] + kp[6 * 3 + c]);
            const double mid_hip = 0.5 * (kp[11 * 3 + c] + kp[12 * 3 + c]);
            dst[51 + c] = 0.5 * (mid_sh + mid_hip);
        }
        double sc = kp[5 * 3 + 2] * kp[6 * 3 + 2];
        sc *= kp[11 * 3 + 2] * kp[12 * 3 + 2];
        dst[53] = sc;
        const double* Pc = Pmats + (size_t)((q / Pmax) % C) * 12;
        for (int e = 0; e < 12; ++e) S.Pm[tid * 12 + e] = Pc[e];
    }
    __syncthreads();
    for (int idx = tid; idx < nv * NOBS * 3; idx += NT) {
        const int v = idx / (NOBS * 3), r = idx - v * NOBS * 3, k = r / 3, c = r - k * 3;
        S.obs[idx] = pose18[(v * 18 + kIkObs[k]) * 3 + c];
    }
    // ---- initial parameters ----
    const bool is_cold = (cold == nullptr) || cold[b] != 0;
    if (is_cold) {
        // root = midpoint of the triangulated (post-optimised) hips; zero angles; reference lengths
        // (inverse_kinematics.py:390-396 with triangulate(..., 0.01, post_optimize=True))
        ik_cold_root(S, pose18, nv);
        for (int i = tid; i < 54; i += NT) S.x[3 + i] = 0.0;
        if (tid < n_side) { S.side[tid] = S.ref_side[tid]; S.x[57 + tid] = S.ref_side[tid]; }
        __syncthreads();
        if (tid < 3) S.x[tid] = 0.5 * (S.xn[tid] + S.xn[3 + tid]);
    } else {
        const double* p0 = init + (size_t)b * 68;
        for (int i = tid; i < 57 + n_side; i += NT) S.x[i] = p0[i];
        if (tid < n_side) S.side[tid] = p0[57 + tid];
    }
    __syncthreads();

    const int max_nfev = is_cold ? nfev_cold : nfev_warm;
    double* fb = scratch + (size_t)b * MVMC_IK_SCRATCH_DOUBLES;
#ifdef MVMC_IK_PROFILE
    if (tid < 8) S.prof[tid] = 0;
    const long long t_all = clock64();
    __syncthreads();
#endif
    // the two stages share one call site (a loop, not two calls): one copy of the solver in the kernel, no
    // call/return spills
    double costs[2];
    int nfs[2], njs[2], sts[2], fallbacks = 0;
#pragma unroll 1
    for (int stage = 0; stage < 2; ++stage) {
        double c; int nf, nj, st;
        ik_trf(S, sk, stage, max_nfev, fb, &c, &nf, &nj, &st, &fallbacks);
        costs[stage] = c; nfs[stage] = nf; njs[stage] = nj; sts[stage] = st;
        __syncthreads();
    }
    const double cost1 = costs[0], cost2 = costs[1];
    const int nf1 = nfs[0], nf2 = nfs[1], nj1 = njs[0], nj2 = njs[1], st1 = sts[0], st2 = sts[1];
    // final FK at the solution
    ik_eval(S, sk, S.x, 1, false);
    __syncthreads();
    for (int i = tid; i < 57 + n_side; i += NT) params_out[(size_t)b * 68 + i] = S.x[i];
    for (int i = tid; i < 54; i += NT) joints_out[(size_t)b * 54 + i] = S.pos[i];
    if (info && tid == 0) {
        info[0] = cost1; info[1] = nf1; info[2] = st1; info[3] = cost2; info[4] = nf2; info[5] = st2;
        info[6] = nj1 + nj2; info[7] = fallbacks;
#ifdef MVMC_IK_PROFILE
        // diagnostic build only: cycle shares instead of the costs
        info[0] = (double)S.prof[0]; info[3] = (double)S.prof[1]; info[2] = (double)S.prof[2];
        info[5] = (double)(clock64() - t_all); info[7] = (double)S.prof[3];
        info[1] = (double)S.prof[4]; info[4] = (double)S.prof[5]; info[6] = (double)S.prof[6];
#endif
    }
}

}  // namespace

// mvmc_ik1.hip
int mvmc_ik1_launch(const SkelDev& sk, const double* kps17, const double* Pmats, const int32_t* members, int n_problems,
                    int v_max, int n_views, int p_max, const double* init_params, const uint8_t* cold, int max_nfev_cold,
                    int max_nfev_warm, double* params_out, double* joints_out, double* info_out, double* scratch,
                    int stage_mask, const double* targets3d, hipStream_t stream);

// 0: wave-per-solve kernel (default); 1: workgroup-per-solve kernel.  MVMC_IK_MODE in the environment sets the initial value.
static int g_ik_mode = -1;
extern "C" int mvmc_debug_ik_mode(int mode) {
    const int prev = g_ik_mode;
    if (mode == 0 || mode == 1) g_ik_mode = mode;
    return prev;
}

extern "C" int mvmc_ik_solve(const mvmcSkeleton* skel_host, const double* kps17, const double* Pmats,
                             const int32_t* members, int n_problems, int v_max, int n_views, int p_max,
                             const double* init_params, const uint8_t* cold, int max_nfev_cold, int max_nfev_warm,
                             double* params_out, double* joints_out, double* info_out, double* scratch,
                             mvmcStream_t stream) {
    if (!skel_host || !kps17 || !Pmats || !members || !params_out || !joints_out || !scratch) return MVMC_ERR_ARG;
    if (v_max <= 0 || n_views <= 0 || p_max <= 0 || max_nfev_cold < 1 || max_nfev_warm < 1) return MVMC_ERR_ARG;
    if (cold && !init_params) return MVMC_ERR_ARG;
    if (n_problems <= 0) return n_problems == 0 ? MVMC_OK : MVMC_ERR_ARG;
    SkelDev sk;
    if (!skel_to_dev(skel_host, &sk)) return MVMC_ERR_ARG;
    if (sk.n_side != MVMC_N_SIDE) return MVMC_ERR_UNSUPPORTED;  // the solver is sized for 57 + 11 parameters
    if (g_ik_mode < 0) {
        const char* e = getenv("MVMC_IK_MODE");
        g_ik_mode = (e && e[0] == '1') ? 1 : 0;
    }
    const uint8_t* cold_arg = init_params ? cold : nullptr;
    if (g_ik_mode == 0)
        return mvmc_ik1_launch(sk, kps17, Pmats, members, n_problems, v_max, n_views, p_max, init_params, cold_arg, max_nfev_cold,
                               max_nfev_warm, params_out, joints_out, info_out, scratch, 3, nullptr, (hipStream_t)stream);
    hipLaunchKernelGGL(ik_kernel, dim3(n_problems), dim3(NT), 0, (hipStream_t)stream, sk, kps17, Pmats, members,
                       n_problems, v_max, n_views, p_max, init_params, cold_arg, max_nfev_cold,
                       max_nfev_warm, params_out, joints_out, info_out, scratch);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}

// Single stages of PoseSolver.solve and the 3-D-target variants (wave-per-solve kernel only); see include/mvmc.h
extern "C" int mvmc_ik_solve_stages(const mvmcSkeleton* skel_host, const double* kps17, const double* Pmats,
                                    const int32_t* members, const double* targets3d, int n_problems, int v_max, int n_views,
                                    int p_max, const double* init_params, int stage_mask, int max_nfev,
                                    double* params_out, double* joints_out, double* info_out, double* scratch,
                                    mvmcStream_t stream) {
    if (!skel_host || !init_params || !params_out || !joints_out || !scratch) return MVMC_ERR_ARG;
    if (stage_mask < 1 || stage_mask > 3 || max_nfev < 1) return MVMC_ERR_ARG;
    if (!targets3d && (!kps17 || !Pmats || !members || v_max <= 0 || n_views <= 0 || p_max <= 0)) return MVMC_ERR_ARG;
    if (n_problems <= 0) return n_problems == 0 ? MVMC_OK : MVMC_ERR_ARG;
    SkelDev sk;
    if (!skel_to_dev(skel_host, &sk)) return MVMC_ERR_ARG;
    if (sk.n_side != MVMC_N_SIDE) return MVMC_ERR_UNSUPPORTED;
    // every problem is "warm": it starts from init_params with the one evaluation budget
    return mvmc_ik1_launch(sk, kps17, Pmats, members, n_problems, targets3d ? 1 : v_max, targets3d ? 1 : n_views,
                           targets3d ? 1 : p_max, init_params, /*cold=*/nullptr, max_nfev, max_nfev, params_out, joints_out,
                           info_out, scratch, stage_mask | 4, targets3d, (hipStream_t)stream);
}
