// match_svt (mv_association.py:321-411): ADMM with singular-value thresholding and the doubly-stochastic block projection
// (myproj2dpam / projR / projC / proj2pav, mv_association.py:15-60), one 256-thread workgroup per graph, n <= 64 nodes.
// Not on the live path of the reference (match_als is what match_spatial / match_spatial_time call); kept for the callers that
// select it.
//
// Per iteration the reference takes torch.svd of M = Y / mu + X.  M is symmetric (X is symmetrised every iteration, Y only ever
// receives symmetric updates), so U diag(max(s - lambda / mu, 0)) V^T = sum_k sign(l_k) max(|l_k| - lambda / mu, 0) v_k v_k^T with the
// eigenpairs (l_k, v_k) of M: a cyclic Jacobi eigensolver in LDS (round-robin pairs, n / 2 rotations per step, three barriers
// per step) replaces the SVD.  Arithmetic is float64 whatever the input type (the reference keeps a float32 S in float32; the
// results agree to float32 rounding, X_bin / match_mat / iteration counts equal on the fixtures of tests/golden/svt_cases.npz).
//
// Memory: LDS holds X, the eigen work matrix A (which then receives Q) and V (eigenvectors; reused for the row-projected blocks);
// Y and the two projection states live in the caller's workspace (3 n_max^2 doubles per graph, touched by this workgroup only).
#include "mvmc_common.h"

namespace {

constexpr int SVT_NMAX = 64, SVT_GMAX = 16, SVT_GROUPS = 16, SVT_NT = 256;

// proj2pav (mv_association.py:49-60) on a vector held in registers: negatives clipped; kept if it sums below 1, else the
// simplex projection max(y - theta, 0).  theta comes from the element of largest descending rank k whose value exceeds
// (prefix sum - 1) / k -- ranks and prefix sums by counting (stable order), no sort and no dynamic register indexing.
template <int GM>
__device__ __forceinline__ void proj2pav_regs(double (&y)[GM], int len) {
    double sum = 0.0;
#pragma unroll
    for (int i = 0; i < GM; ++i) {
        if (i < len) { if (y[i] < 0.0) y[i] = 0.0; sum += y[i]; }
        else y[i] = 0.0;
    }
    if (sum < 1.0) return;
    int best_k = 0;
    double best_s = 0.0;
#pragma unroll
    for (int i = 0; i < GM; ++i) {
        if (i < len) {
            int k = 0;
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < GM; ++j)
                if (j < len && (y[j] > y[i] || (y[j] == y[i] && j <= i))) { ++k; s += y[j]; }
            if (y[i] > (s - 1.0) / k && k > best_k) { best_k = k; best_s = s; }
        }
    }
    double theta = (best_s - 1.0) / best_k;   // the largest element always qualifies (u0 > u0 - 1), so best_k >= 1
    if (theta < 0.0) theta = 0.0;
#pragma unroll
    for (int i = 0; i < GM; ++i)
        if (i < len) { const double v = y[i] - theta; y[i] = v > 0.0 ? v : 0.0; }
}

__device__ __forceinline__ double block_sum(double v, double* red) {   // deterministic: DPP wave sums, four parts in order
    v = wave_sum_dpp(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

template <typename TS, int GM>
__global__ void __launch_bounds__(SVT_NT)
svt_kernel(const TS* __restrict__ Sg, const int32_t* __restrict__ gcounts, int G, int ldw, double alpha, double lam, double mu0,
           double tol, int max_iter, int dual, double* __restrict__ work, uint8_t* __restrict__ x_bin, double* __restrict__ x_out,
           int32_t* __restrict__ iters_out) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ int dimg[SVT_GROUPS + 1];
    __shared__ int grp_of[SVT_NMAX];
    __shared__ double rot_c[SVT_NMAX / 2], rot_s[SVT_NMAX / 2], rot_off[SVT_NMAX / 2], fk[SVT_NMAX];
    __shared__ int rot_p[SVT_NMAX / 2], rot_q[SVT_NMAX / 2];
    __shared__ double partc[SVT_NMAX * SVT_GROUPS];
    __shared__ int blk_done[SVT_GROUPS * SVT_GROUPS];
    __shared__ int s_flag;
    __shared__ double red[4];
    const int tid = threadIdx.x, f = blockIdx.x;
    if (tid == 0) {
        int acc = 0;
        dimg[0] = 0;
        for (int g = 0; g < G; ++g) { acc += gcounts[(size_t)f * G + g]; dimg[g + 1] = acc; }
    }
    __syncthreads();
    const int n = dimg[G];
    const int m = (n + 1) & ~1;   // even size for the Jacobi pairing (the pad row / column is zero: an eigenvalue 0, untouched)
    double* X = lds;
    double* A = lds + m * m;
    double* V = lds + 2 * m * m;
    const TS* S = Sg + (size_t)f * ldw * ldw;
    double* Yg = work + (size_t)f * 3 * ldw * ldw;
    double* Pg = Yg + ldw * ldw;
    double* Cg = Pg + ldw * ldw;
    for (int i = tid; i < n; i += SVT_NT) {
        int g = 0;
        while (g + 1 < G && i >= dimg[g + 1]) ++g;
        grp_of[i] = g;
    }
    // S <- (S + S^T) / 2 with a zero diagonal; X = S; Y = 0
    auto s_sym = [&](int i, int j) { return i == j ? 0.0 : 0.5 * ((double)S[i * ldw + j] + (double)S[j * ldw + i]); };
    for (int e = tid; e < m * m; e += SVT_NT) {
        const int i = e / m, j = e - i * m;
        const bool in = i < n && j < n;
        X[e] = in ? s_sym(i, j) : 0.0;
        if (in) Yg[i * ldw + j] = 0.0;
    }
    __syncthreads();
    if (n == 0) {
        if (tid == 0) iters_out[f] = 0;
        return;   // (uniform)
    }
    double mu = mu0;
    int n_run = max_iter;
    constexpr int EPT = SVT_NMAX * SVT_NMAX / SVT_NT;   // matrix entries per thread
    for (int it = 0; it < max_iter; ++it) {
        const double inv_mu = 1.0 / mu;
        double xprev[EPT];
        // ---- M = Y / mu + X into A; V = I ----
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            const int e = tid + u * SVT_NT;
            xprev[u] = 0.0;
            if (e < m * m) {
                const int i = e / m, j = e - i * m;
                const double x = X[e];
                xprev[u] = x;
                A[e] = (i < n && j < n) ? inv_mu * Yg[i * ldw + j] + x : 0.0;
                V[e] = i == j ? 1.0 : 0.0;
            }
        }
        __syncthreads();
        // ---- cyclic Jacobi: A <- J^T A J, V <- V J until the off-diagonal part is at rounding level ----
        const int half = m / 2;
        for (int sweep = 0; sweep < 30; ++sweep) {
            double off_max = 0.0, diag_max = 0.0;
            for (int step = 0; step < m - 1; ++step) {
                if (tid < half) {
                    // round-robin pairing of m players: player m-1 stays, the others rotate
                    int p, q;
                    if (tid == 0) { p = m - 1; q = step; }
                    else { p = (step + tid) % (m - 1); q = (step - tid + (m - 1)) % (m - 1); }
                    if (p > q) { const int t = p; p = q; q = t; }
                    const double app = A[p * m + p], aqq = A[q * m + q], apq = A[p * m + q];
                    double c = 1.0, s = 0.0;
                    if (apq != 0.0) {
                        const double th = (aqq - app) / (2.0 * apq);
                        const double t = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                        c = 1.0 / sqrt(t * t + 1.0);
                        s = t * c;
                    }
                    rot_p[tid] = p; rot_q[tid] = q; rot_c[tid] = c; rot_s[tid] = s;
                    off_max = fmax(off_max, fabs(apq));
                    diag_max = fmax(diag_max, fmax(fabs(app), fabs(aqq)));
                }
                __syncthreads();
                for (int e = tid; e < m * half; e += SVT_NT) {   // columns p, q of A and V
                    const int i = e / half, t = e - i * half;
                    const int p = rot_p[t], q = rot_q[t];
                    const double c = rot_c[t], s = rot_s[t];
                    const double aip = A[i * m + p], aiq = A[i * m + q];
                    A[i * m + p] = c * aip - s * aiq;
                    A[i * m + q] = s * aip + c * aiq;
                    const double vip = V[i * m + p], viq = V[i * m + q];
                    V[i * m + p] = c * vip - s * viq;
                    V[i * m + q] = s * vip + c * viq;
                }
                __syncthreads();
                for (int e = tid; e < m * half; e += SVT_NT) {   // rows p, q of A
                    const int t = e / m, j = e - t * m;
                    const int p = rot_p[t], q = rot_q[t];
                    const double c = rot_c[t], s = rot_s[t];
                    const double apj = A[p * m + j], aqj = A[q * m + j];
                    A[p * m + j] = c * apj - s * aqj;
                    A[q * m + j] = s * apj + c * aqj;
                }
                __syncthreads();
            }
            if (tid < half) { rot_off[tid] = off_max; rot_c[tid] = diag_max; }
            __syncthreads();
            double om = 0.0, dm = 0.0;
            for (int t = 0; t < half; ++t) { om = fmax(om, rot_off[t]); dm = fmax(dm, rot_c[t]); }
            __syncthreads();
            if (om <= 1e-15 * dm || om == 0.0) break;   // (uniform)
        }
        // ---- Q = sum_k sign(l_k) max(|l_k| - lambda / mu, 0) v_k v_k^T, into A ----
        if (tid < m) {
            const double l = A[tid * m + tid];
            const double sh = fabs(l) - lam * inv_mu;
            fk[tid] = sh > 0.0 ? (l >= 0.0 ? sh : -sh) : 0.0;
        }
        __syncthreads();
        for (int e = tid; e < m * m; e += SVT_NT) {
            const int i = e / m, j = e - i * m;
            double q = 0.0;
            for (int k = 0; k < m; ++k) q += V[i * m + k] * fk[k] * V[j * m + k];
            A[e] = q;
        }
        __syncthreads();
        // ---- X = Q - (W + Y) / mu, W = alpha - S; same-group blocks 0, diagonal 1, clipped to [0, 1] ----
        for (int e = tid; e < m * m; e += SVT_NT) {
            const int i = e / m, j = e - i * m;
            double x = 0.0;
            if (i < n && j < n) {
                x = A[e] - ((alpha - s_sym(i, j)) + Yg[i * ldw + j]) * inv_mu;
                if (grp_of[i] == grp_of[j]) x = 0.0;
                if (i == j) x = 1.0;
                x = x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x);
            }
            X[e] = x;
        }
        __syncthreads();
        if (dual) {
            // ---- every (row group, column group) block onto the doubly-stochastic set: Dykstra's alternating row / column
            //      projections, at most 10 rounds per block, a block stops when its mean change drops below 1e-2.
            //      State per entry: P (the column correction I2), C (the current iterate); the row-projected matrix goes to V.
            //      X0 + I1 = X1 - I2 (I1 = X1 - (X0 + I2)), so the row correction needs no storage. ----
            for (int e = tid; e < n * n; e += SVT_NT) {
                const int i = e / n, j = e - i * n;
                Pg[i * ldw + j] = 0.0;
                Cg[i * ldw + j] = X[i * m + j];
            }
            for (int e = tid; e < G * G; e += SVT_NT) {
                const int gi = e / G, gj = e - gi * G;
                blk_done[e] = (dimg[gi + 1] == dimg[gi] || dimg[gj + 1] == dimg[gj]) ? 1 : 0;
            }
            __syncthreads();
            for (int round = 0; round < 10; ++round) {
                for (int e = tid; e < n * G; e += SVT_NT) {   // rows: task (row i, column group gj)
                    const int i = e / G, gj = e - i * G;
                    if (blk_done[grp_of[i] * G + gj]) continue;
                    const int c0 = dimg[gj], len = dimg[gj + 1] - c0;
                    double y[GM];
#pragma unroll
                    for (int c = 0; c < GM; ++c) y[c] = c < len ? X[i * m + c0 + c] + Pg[i * ldw + c0 + c] : 0.0;
                    proj2pav_regs<GM>(y, len);
#pragma unroll
                    for (int c = 0; c < GM; ++c)
                        if (c < len) V[i * m + c0 + c] = y[c];
                }
                __syncthreads();
                for (int e = tid; e < n * G; e += SVT_NT) {   // columns: task (column j, row group gi)
                    const int j = e / G, gi = e - j * G;
                    double part = 0.0;
                    if (!blk_done[gi * G + grp_of[j]]) {
                        const int r0 = dimg[gi], len = dimg[gi + 1] - r0;
                        double y[GM], t[GM];
#pragma unroll
                        for (int r = 0; r < GM; ++r) { t[r] = r < len ? V[(r0 + r) * m + j] - Pg[(r0 + r) * ldw + j] : 0.0; y[r] = t[r]; }
                        proj2pav_regs<GM>(y, len);
#pragma unroll
                        for (int r = 0; r < GM; ++r)
                            if (r < len) {
                                const int o = (r0 + r) * ldw + j;
                                Pg[o] = y[r] - t[r];
                                part += fabs(y[r] - Cg[o]);
                                Cg[o] = y[r];
                            }
                    }
                    partc[j * G + gi] = part;
                }
                __syncthreads();
                if (tid == 0) s_flag = 0;
                __syncthreads();
                for (int e = tid; e < G * G; e += SVT_NT) {
                    const int gi = e / G, gj = e - gi * G;
                    if (!blk_done[e]) {
                        double s = 0.0;
                        for (int j = dimg[gj]; j < dimg[gj + 1]; ++j) s += partc[j * G + gi];
                        const double chg = s / ((dimg[gi + 1] - dimg[gi]) * (dimg[gj + 1] - dimg[gj]));
                        if (chg < 1e-2) blk_done[e] = 1;
                        else atomicOr(&s_flag, 1);
                    }
                }
                __syncthreads();
                if (!s_flag) break;   // (uniform)
            }
            __syncthreads();
            for (int e = tid; e < n * n; e += SVT_NT) {
                const int i = e / n, j = e - i * n;
                X[i * m + j] = Cg[i * ldw + j];
            }
            __syncthreads();
        }
        // ---- X <- (X + X^T) / 2; Y += mu (X - Q); residuals ----
        for (int e = tid; e < n * n; e += SVT_NT) {
            const int i = e / n, j = e - i * n;
            if (i <= j) {
                const double v = 0.5 * (X[i * m + j] + X[j * m + i]);
                X[i * m + j] = v;
                X[j * m + i] = v;
            }
        }
        __syncthreads();
        double sp = 0.0, sd = 0.0;
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            const int e = tid + u * SVT_NT;
            if (e < m * m) {
                const int i = e / m, j = e - i * m;
                if (i < n && j < n) {
                    const double x = X[e], q = A[e];
                    Yg[i * ldw + j] += mu * (x - q);
                    sp += (x - q) * (x - q);
                    sd += (x - xprev[u]) * (x - xprev[u]);
                }
            }
        }
        sp = block_sum(sp, red);
        sd = block_sum(sd, red);
        const double p_res = sqrt(sp) / n, d_res = mu * sqrt(sd) / n;
        __syncthreads();
        if (p_res < tol && d_res < tol) { n_run = it + 1; break; }   // (uniform)
        if (p_res > 10 * d_res) mu = 2 * mu;
        else if (d_res > 10 * p_res) mu = mu / 2;
    }
    // X_bin = X > 0.5 (X is symmetric already), zero outside the n x n block
    for (int e = tid; e < ldw * ldw; e += SVT_NT) {
        const int i = e / ldw, j = e - i * ldw;
        const bool in = i < n && j < n;
        const double x = in ? X[i * m + j] : 0.0;
        x_bin[(size_t)f * ldw * ldw + e] = in && x > 0.5;
        if (x_out) x_out[(size_t)f * ldw * ldw + e] = x;
    }
    if (tid == 0) iters_out[f] = n_run;
}

template <typename TS>
int launch_svt(const TS* S, const int32_t* gcounts, int B, int G, int ldw, int g_max, double alpha, double lam, double mu0, double tol,
               int max_iter, int dual, double* work, uint8_t* x_bin, double* x_out, int32_t* iters, hipStream_t s) {
    const int m = (ldw + 1) & ~1;
    const size_t lds = (size_t)3 * m * m * sizeof(double);
    auto go = [&](auto kern) -> int {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return MVMC_ERR_LAUNCH;
        hipLaunchKernelGGL(kern, dim3(B), dim3(SVT_NT), lds, s, S, gcounts, G, ldw, alpha, lam, mu0, tol, max_iter, dual, work, x_bin,
                           x_out, iters);
        return MVMC_OK;
    };
    return g_max <= 8 ? go(svt_kernel<TS, 8>) : go(svt_kernel<TS, SVT_GMAX>);
}

}  // namespace

extern "C" int mvmc_svt_associate(const void* S, int s_dtype, const int32_t* group_counts, int n_frames, int n_groups, int n_max,
                                  int g_max, double alpha, double lambda, double mu, double tol, int max_iter, int dual_stochastic,
                                  double* work, uint8_t* x_bin, double* x_out, int32_t* iters, mvmcStream_t stream) {
    if (!S || !group_counts || !work || !x_bin || !iters) return MVMC_ERR_ARG;
    if (s_dtype != MVMC_F32 && s_dtype != MVMC_F64) return MVMC_ERR_ARG;
    if (n_frames < 0 || n_groups <= 0 || n_groups > SVT_GROUPS || n_max <= 0 || n_max > SVT_NMAX) return MVMC_ERR_ARG;
    if (g_max <= 0 || g_max > SVT_GMAX || max_iter < 0 || !(mu > 0.0)) return MVMC_ERR_ARG;
    if (n_frames == 0) return MVMC_OK;
    hipStream_t s = (hipStream_t)stream;
    const int st = s_dtype == MVMC_F32
                       ? launch_svt<float>((const float*)S, group_counts, n_frames, n_groups, n_max, g_max, alpha, lambda, mu, tol,
                                           max_iter, dual_stochastic, work, x_bin, x_out, iters, s)
                       : launch_svt<double>((const double*)S, group_counts, n_frames, n_groups, n_max, g_max, alpha, lambda, mu, tol,
                                            max_iter, dual_stochastic, work, x_bin, x_out, iters, s);
    if (st != MVMC_OK) return st;
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}
