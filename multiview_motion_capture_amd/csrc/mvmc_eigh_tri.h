// Symmetric eigensolver for one small (n <= 50) matrix per 256-thread workgroup, everything in LDS:
//   1. Householder tridiagonalisation  A = Q T Q^T          (n-2 steps, 3 barriers each)
//   2. eigenvalues of T by 4-way multisection of Sturm counts (one lane per eigenvalue, one probe per wave)
//   3. eigenvectors of T by twisted factorisation (forward / backward pivots on two waves)
//   4. back-transformation z -> Q z with the eigenvector segments held in registers (no barriers)
//   5. the numerically-null cluster (lambda <= 1e-13 lambda_max) is not resolved into vectors (zero rows)
// Replaces the cyclic Jacobi of the first version (~320 LDS-bound steps) with ~50 cheaper steps.
#pragma once
#include "mvmc_common.h"

namespace eightri {

constexpr int NTH = 256;

__device__ inline double bsum(double v, double* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// number of eigenvalues of T (diag d, squared off-diagonals e2) that are < x; product form of the
// Sturm sequence with rescaling (no divisions)
// Sturm count: eigenvalues of T below x = sign changes along p_j = (d_j - x) p_{j-1} - e_{j-1}^2 p_{j-2}.
// Per step: one add, one mul, one FMA and one integer op (the sign bit is shifted into a history word with
// v_alignbit; sign changes are counted once per block of 8 with a popcount).  dpad / e2pad are padded to whole
// blocks with neutral steps (d = 4 > any x of the scaled matrix, e^2 = 0: p_j = (4 - x) p_{j-1} keeps its sign),
// so the loop has no predication; the (wave-uniform) coefficients are loaded a block at a time.
__device__ inline int sturm_count(const double* dpad, const double* e2pad, int nblk, double x) {
    double pm = 1.0, p = dpad[0] - x;
    unsigned hist = (unsigned)__double2hiint(p) >> 31;  // bit 0 = sign of p_0 (p_{-1} = 1 is positive)
    int cnt = (int)hist;
    for (int b = 0; b < nblk; ++b) {
        double dj[8], ej[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { dj[u] = dpad[1 + 8 * b + u]; ej[u] = e2pad[8 * b + u]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double pn = (dj[u] - x) * p - ej[u] * pm;
            hist = __builtin_amdgcn_alignbit(hist, (unsigned)__double2hiint(pn), 31);  // (hist << 1) | sign(pn)
            pm = p; p = pn;
        }
        // bits 8..0 of hist = signs of the block's 9 consecutive terms (oldest = the last term of the previous block)
        cnt += __popc((hist ^ (hist >> 1)) & 0xffu);
        const double a = fmax(fabs(p), fabs(pm));
        if (a > 1e100) { p *= 1e-100; pm *= 1e-100; }
        else if (a < 1e-100) { p *= 1e100; pm *= 1e100; }
    }
    return cnt;
}

// A (n x n, ld lda, full symmetric, destroyed) -> lam (n, ascending, null cluster set to 0),
// Zt (rows = eigenvectors, ld ldz; rows 0..k0-1 of the null cluster are zero).
// W1: n x ldw scratch (>= 128 doubles).  d, e, tau, pv, wv: LDS vectors of >= n doubles.  icnt: 16*64 ints.
// Returns k0.
__device__ int eigh(double* A, int lda, double* Zt, int ldz, double* W1, int ldw, int n,
                    double* lam, double* d, double* e, double* tau, double* pv, double* wv, double* red, int* icnt,
                    long long* prof = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wv_id = tid >> 6;
    long long t_prev = prof ? clock64() : 0;
    auto stamp = [&](int k) { if (prof) { const long long t = clock64(); if (tid == 0) prof[k] = t - t_prev; t_prev = t; } };
    // ---------------- 1. tridiagonalisation (lower form: column k holds v_k below the sub-diagonal) ----------------
    // sv = pv (Householder vector, contiguous copy), sw = wv
    double* sv = pv;
    for (int k = 0; k < n - 2; ++k) {
        const int m = n - k - 1;
        if (tid < 64) {
            const double xi = (lane < m) ? A[(k + 1 + lane) * lda + k] : 0.0;
            const double alpha = __shfl(xi, 0, 64);
            const double sig = wave_sum_dpp((lane >= 1 && lane < m) ? xi * xi : 0.0);
            double tk = 0.0, beta = alpha, vi = (lane == 0) ? 1.0 : 0.0;
            if (sig > 0.0) {
                const double q = alpha * alpha + sig;
                double rs = __builtin_amdgcn_rsq(q);
                rs = rs * (1.5 - 0.5 * q * rs * rs);
                rs = rs * (1.5 - 0.5 * q * rs * rs);
                const double nrm = q * rs;
                beta = alpha >= 0.0 ? -nrm : nrm;
                tk = 1.0 - alpha * fast_rcp64(beta);
                const double sc = fast_rcp64(alpha - beta);
                vi = (lane == 0) ? 1.0 : xi * sc;
            }
            if (lane < m) { A[(k + 1 + lane) * lda + k] = vi; sv[lane] = vi; }
            if (lane == 0) { e[k] = beta; tau[k] = tk; }
        }
        __syncthreads();
        const double tk = tau[k];
        if (tk != 0.0) {
            // p = tau * A22 v, four lanes per row, all loads of a lane issued before the sums
            {
                const int r = tid >> 2, qd = tid & 3;
                double s = 0.0;
                if (r < m) {
                    const double* row = &A[(k + 1 + r) * lda + (k + 1)];
                    double a[13], b[13];
#pragma unroll
                    for (int t = 0; t < 13; ++t) {
                        const int j = qd + 4 * t;
                        const int jj = j < m ? j : 0;
                        a[t] = row[jj]; b[t] = j < m ? sv[jj] : 0.0;
                    }
#pragma unroll
                    for (int t = 0; t < 13; ++t) s += a[t] * b[t];
                }
                s = quad_sum(s);
                if (qd == 0 && r < m) wv[r] = tk * s;   // p
            }
            __syncthreads();
            if (tid < 64) {
                const double vi = (lane < m) ? sv[lane] : 0.0;
                const double pi = (lane < m) ? wv[lane] : 0.0;
                const double dot = wave_sum_dpp(pi * vi);
                if (lane < m) wv[lane] = pi - 0.5 * tk * dot * vi;   // w (in place: only this wave touches wv here)
            }
            __syncthreads();
            // A22 -= v w^T + w v^T on a 16 x 16 thread grid
            {
                const int ty = tid >> 4, tx = tid & 15;
                double vj[4], wj[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = tx + 16 * u;
                    vj[u] = j < m ? sv[j] : 0.0; wj[u] = j < m ? wv[j] : 0.0;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = ty + 16 * q;
                    if (i < m) {
                        const double vi = sv[i], wi = wv[i];
                        double* row = &A[(k + 1 + i) * lda + (k + 1)];
                        double cur[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const int j = tx + 16 * u; cur[u] = j < m ? row[j] : 0.0; }
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const int j = tx + 16 * u; if (j < m) row[j] = cur[u] - (vi * wj[u] + wi * vj[u]); }
                    }
                }
            }
            __syncthreads();
        }
    }
    stamp(0);
    if (tid < n) d[tid] = A[tid * lda + tid];
    if (tid == 0) { e[n - 2] = A[(n - 1) * lda + (n - 2)]; tau[n - 2] = 0.0; }
    __syncthreads();
    // squared off-diagonals in pv
    if (tid < n - 1) pv[tid] = e[tid] * e[tid];
    __syncthreads();
    // ---------------- 2. eigenvalues: Gershgorin interval + 4-way multisection ----------------
    double lo, hi, tscale;
    {
        double gl = 1e300, gu = -1e300;
        if (lane < n) {
            const double r = (lane > 0 ? fabs(e[lane - 1]) : 0.0) + (lane < n - 1 ? fabs(e[lane]) : 0.0);
            gl = d[lane] - r; gu = d[lane] + r;
        }
        for (int off = 32; off > 0; off >>= 1) { gl = fmin(gl, __shfl_xor(gl, off, 64)); gu = fmax(gu, __shfl_xor(gu, off, 64)); }
        const double bn = fmax(fabs(gl), fabs(gu));
        tscale = bn > 0.0 ? bn : 1.0;
        lo = (gl - 2.2e-14 * bn) / tscale - 1e-290; hi = (gu + 2.2e-14 * bn) / tscale + 1e-290;
    }
    // Sturm counts run on T / tscale (|entries| <= 1), padded to whole blocks of 8 with neutral steps;
    // the scratch matrix W1 is free until the twisted factorisation: dsc = W1[0..64), e2sc = W1[64..128)
    double* dsc = W1;
    double* e2sc = W1 + 64;
    __syncthreads();
    if (tid < 64) {
        dsc[tid] = tid < n ? d[tid] / tscale : 4.0;
        const double es = tid < n - 1 ? e[tid] / tscale : 0.0;
        e2sc[tid] = es * es;
    }
    __syncthreads();
    const int nblk = (n - 1 + 7) >> 3;
    // one probe per wave: 4-way multisection, the bracket shrinks 5x per round
    for (int round = 0; round < 26; ++round) {
        const double w = (hi - lo) * 0.2;
        const int c = sturm_count(dsc, e2sc, nblk, lo + w * (wv_id + 1));
        if (lane < n) icnt[wv_id * 64 + lane] = c;
        __syncthreads();
        if (lane < n) {
            double nlo = lo, nhi = hi;
            bool found = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double xq = lo + w * (q + 1);
                if (!found) {
                    if (icnt[q * 64 + lane] > lane) { nhi = xq; found = true; }
                    else nlo = xq;
                }
            }
            lo = nlo; hi = nhi;
        }
        __syncthreads();
    }
    if (tid < n) lam[tid] = 0.5 * (lo + hi) * tscale;
    if (tid < n - 1) pv[tid] = e[tid] * e[tid];  // unscaled squares for the twisted factorisation
    __syncthreads();
    stamp(1);
    const double lmax = fmax(fabs(lam[n - 1]), fabs(lam[0]));
    const double tol0 = 1e-13 * lmax;
    int k0 = 0;
    for (int i = 0; i < n; ++i) k0 += lam[i] <= tol0;  // ascending: the null cluster is lam[0..k0)
    const double pivmin = 1e-16 * lmax + 1e-300;
    // ---------------- 3. eigenvectors of T: twisted factorisation ----------------
    if (wv_id == 0 && lane < n && lane >= k0) {        // forward pivots D+ -> W1 row
        const double l = lam[lane];
        double dp = d[0] - l;
        for (int j = 0; j < n - 1; ++j) {
            if (fabs(dp) < pivmin) dp = dp < 0.0 ? -pivmin : pivmin;
            W1[lane * ldw + j] = dp;
            dp = (d[j + 1] - l) - pv[j] * fast_rcp64(dp);
        }
        W1[lane * ldw + n - 1] = dp;
    } else if (wv_id == 1 && lane < n && lane >= k0) {  // backward pivots D- -> Zt row
        const double l = lam[lane];
        double dm = d[n - 1] - l;
        for (int j = n - 1; j > 0; --j) {
            if (fabs(dm) < pivmin) dm = dm < 0.0 ? -pivmin : pivmin;
            Zt[lane * ldz + j] = dm;
            dm = (d[j - 1] - l) - pv[j - 1] * fast_rcp64(dm);
        }
        Zt[lane * ldz] = dm;
    }
    __syncthreads();
    if (wv_id == 0 && lane < n && lane >= k0) {
        const double l = lam[lane];
        double* zr = &Zt[lane * ldz];
        const double* dpr = &W1[lane * ldw];
        int r = 0;
        double gmin = 1e300;
        for (int j = 0; j < n; ++j) {
            const double gam = fabs(dpr[j] + zr[j] - (d[j] - l));
            if (gam < gmin) { gmin = gam; r = j; }
        }
        // z_r = 1; upward with U_j = e_j / D-_{j+1}; downward with L_j = e_j / D+_j
        double z = 1.0, nrm = 1.0;
        for (int j = r; j < n - 1; ++j) {
            double dm = zr[j + 1];
            if (fabs(dm) < pivmin) dm = dm < 0.0 ? -pivmin : pivmin;
            z = -e[j] * z * fast_rcp64(dm);
            zr[j + 1] = z;
            nrm += z * z;
        }
        z = 1.0;
        for (int j = r - 1; j >= 0; --j) {
            double dp = dpr[j];
            if (fabs(dp) < pivmin) dp = dp < 0.0 ? -pivmin : pivmin;
            z = -e[j] * z * fast_rcp64(dp);
            zr[j] = z;
            nrm += z * z;
        }
        zr[r] = 1.0;
        const double inv = 1.0 / sqrt(nrm);
        for (int j = 0; j < n; ++j) zr[j] *= inv;
    }
    __syncthreads();
    stamp(2);
    // ---------------- 3b. re-orthogonalise runs of close eigenvalues (gap < 1e-7 lam_max) ----------------
    // Eigenvalues are known to eps*lam_max absolute, so vectors of eigenvalues closer than ~1e-7 lam_max
    // come out only ~1e-9/gap orthogonal; modified Gram-Schmidt inside each run restores it.
    {
        const double gtol = 1e-7 * lmax;
        int start = k0;
        for (int i = k0 + 1; i < n; ++i) {  // uniform control flow: lam is shared
            if (lam[i] - lam[i - 1] >= gtol) { start = i; continue; }
            // row i against rows start..i-1: four lanes per earlier row
            {
                const int q = start + (tid >> 2), qd = tid & 3;
                double s = 0.0;
                if (q < i)
                    for (int j = qd; j < n; j += 4) s += Zt[q * ldz + j] * Zt[i * ldz + j];
                s += __shfl_xor(s, 1, 64);
                s += __shfl_xor(s, 2, 64);
                if (qd == 0 && q < i) wv[q - start] = s;
            }
            __syncthreads();
            double zj = 0.0;
            if (tid < n) {
                zj = Zt[i * ldz + tid];
                for (int q = start; q < i; ++q) zj -= wv[q - start] * Zt[q * ldz + tid];
            }
            const double nn = bsum(zj * zj, red);
            if (tid < n) Zt[i * ldz + tid] = zj / sqrt(nn);
            __syncthreads();
        }
    }
    stamp(3);
    // ---------------- 4. back-transformation: rows of Zt <- (Q z)^T, four lanes per eigenvector ----------------
    {
        const int i = tid >> 2, qd = tid & 3;
        constexpr int TMAX = 13;  // ceil(50 / 4)
        const bool on = i < n && i >= k0;
        double zr[TMAX];
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            const int j = qd + 4 * t;
            zr[t] = (on && j < n) ? Zt[i * ldz + j] : 0.0;
        }
        for (int k = n - 3; k >= 0; --k) {
            const double tk = tau[k];
            if (tk == 0.0) continue;
            double vk[TMAX];
            double s = 0.0;
#pragma unroll
            for (int t = 0; t < TMAX; ++t) {
                const int j = qd + 4 * t;
                vk[t] = (j >= k + 1 && j < n) ? A[j * lda + k] : 0.0;
                s += vk[t] * zr[t];
            }
            s = quad_sum(s) * tk;
#pragma unroll
            for (int t = 0; t < TMAX; ++t) zr[t] -= s * vk[t];
        }
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            const int j = qd + 4 * t;
            if (on && j < n) Zt[i * ldz + j] = zr[t];
        }
    }
    __syncthreads();
    stamp(4);
    // ---------------- 5. the numerically-null cluster is not resolved: zero rows, zero eigenvalues ----------------
    // (g = J^T f has no component there beyond rounding; the trust-region solve models the reference's
    //  noise-filled null directions with one virtual absorber instead, see ik_tr_solve)
    for (int idx = tid; idx < k0 * n; idx += NTH) Zt[(idx / n) * ldz + idx % n] = 0.0;
    if (tid < k0) lam[tid] = 0.0;
    __syncthreads();
    return k0;
}

}  // namespace eightri
