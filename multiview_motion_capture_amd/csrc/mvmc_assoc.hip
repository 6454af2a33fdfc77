// Cross-view association kernels for gfx950 (wave64):
//   ingest     IN-1/IN-2  OpenPose-25 -> COCO-17 gather + filter_bad_pose + per-view compaction
//   fmats      AS-1       pairwise fundamental matrices (f64 math, f32 store)
//   affinity   AS-2/AS-3  epipolar point-line distance blocks + f32 z-score sigmoid
//   als        AS-4/5/6   low-rank ADMM/ALS matching, closure, cluster labels
//   members               labels -> per-cluster member lists
// One frame per workgroup; the frame's whole working set lives in LDS / VGPRs, HBM is touched
// once on the way in and once on the way out (the path is ALU/latency bound).
// As a translation unit of its own (the stand-alone kernels: MVMC_DEVICE_ONLY is what the chain kernels' units define before they include
// this file) it is built for 128 VGPRs and FOUR workgroups per CU like the chain kernel's SMALL layout: als4_kernel -- config 3's
// dominant kernel, a workgroup per graph -- keeps 1,024 graphs resident instead of 768: 844 k -> 905 k frames/s on config 3 (same box),
// the same als7 code the chain kernel runs, results unchanged.
#if !defined(MVMC_DEVICE_ONLY) && !defined(MVMC_SMALL_WPS)
#define MVMC_SMALL_WPS 4
#define ALS4_WG_PER_CU 4
#endif
#include "mvmc_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// ingest
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
ingest_kernel(const T* __restrict__ kps, int C, int P, int J_in, const int32_t* __restrict__ counts_in,
              double min_score, int min_valid, double min_bb, double* __restrict__ kps17,
              int32_t* __restrict__ counts_out) {
    extern __shared__ double sm[];
    const int f = blockIdx.x;
    const int nq = C * P;
    double* pose = sm;                                // [nq][17][3]
    int* keep = reinterpret_cast<int*>(sm + nq * 51);  // [nq]
    int* src_of = keep + nq;                          // [nq] destination slot -> source pose
    const T* src = kps + (size_t)f * nq * J_in * 3;
    for (int e = threadIdx.x; e < nq * 51; e += blockDim.x) {
        int q = e / 51, r = e - q * 51, j = r / 3, k = r - j * 3;
        int js = (J_in == 25) ? op25_to_coco17(j) : j;
        pose[e] = (double)src[(q * J_in + js) * 3 + k];
    }
    __syncthreads();
    for (int q = threadIdx.x; q < nq; q += blockDim.x) {
        int c = q / P, p = q - c * P;
        int cnt = counts_in ? counts_in[f * C + c] : P;
        int ok = 0;
        if (p < cnt) {
            int nv = 0;
            double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
            for (int j = 0; j < 17; ++j) {
                const double* kp = pose + q * 51 + j * 3;
                if (kp[2] > min_score) {
                    ++nv;
                    x0 = fmin(x0, kp[0]); x1 = fmax(x1, kp[0]);
                    y0 = fmin(y0, kp[1]); y1 = fmax(y1, kp[1]);
                }
            }
            ok = (nv >= min_valid) && !((x1 - x0) < min_bb || (y1 - y0) < min_bb);
        }
        keep[q] = ok;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        int k = 0;
        for (int p = 0; p < P; ++p)
            if (keep[c * P + p]) src_of[c * P + k++] = c * P + p;
        counts_out[f * C + c] = k;
        for (; k < P; ++k) src_of[c * P + k] = -1;
    }
    __syncthreads();
    double* out = kps17 + (size_t)f * nq * 51;
    for (int e = threadIdx.x; e < nq * 51; e += blockDim.x) {
        int d = e / 51, r = e - d * 51;
        int q = src_of[d];
        out[e] = q >= 0 ? pose[q * 51 + r] : 0.0;
    }
}

// ------------------------------------------------------------------------------------------------
// fundamental matrices
// ------------------------------------------------------------------------------------------------
__device__ inline void m3mul(const double* A, const double* B, double* O) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * B[k * 3 + j];
            O[i * 3 + j] = s;
        }
}
__device__ inline void m3t(const double* A, double* O) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) O[i * 3 + j] = A[j * 3 + i];
}
__device__ inline void m3v(const double* A, const double* x, double* o) {
    for (int i = 0; i < 3; ++i) o[i] = A[i * 3] * x[0] + A[i * 3 + 1] * x[1] + A[i * 3 + 2] * x[2];
}
__device__ inline void m3inv(const double* A, double* O) {
    double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
    double det = A[0] * c00 + A[1] * c01 + A[2] * c02, id = 1.0 / det;
    O[0] = c00 * id; O[1] = (A[2] * A[7] - A[1] * A[8]) * id; O[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    O[3] = c01 * id; O[4] = (A[0] * A[8] - A[2] * A[6]) * id; O[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    O[6] = c02 * id; O[7] = (A[1] * A[6] - A[0] * A[7]) * id; O[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}

// F_ij = inv(K_i)^T (R_i R_j^T) K_j^T [K_j R_j R_i^T (t_i - R_i R_j^T t_j)]_x   (mv_math_util.py:268-283)
__global__ void fmats_kernel(const double* __restrict__ K, const double* __restrict__ Rt, int C,
                             float* __restrict__ F) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= C * C) return;
    int i = idx / C, j = idx - i * C;
    double K0[9], K1[9], R0[9], R1[9], T0[3], T1[3];
    for (int a = 0; a < 9; ++a) { K0[a] = K[i * 9 + a]; K1[a] = K[j * 9 + a]; }
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) { R0[a * 3 + b] = Rt[i * 12 + a * 4 + b]; R1[a * 3 + b] = Rt[j * 12 + a * 4 + b]; }
        T0[a] = Rt[i * 12 + a * 4 + 3]; T1[a] = Rt[j * 12 + a * 4 + 3];
    }
    double Ki[9], KiT[9], R1T[9], R0T[9], K1T[9], R01[9], A[9], B[9], Cm[9], D[9], v[3], w[3], u[3];
    m3inv(K0, Ki); m3t(Ki, KiT); m3t(R1, R1T); m3t(R0, R0T); m3t(K1, K1T);
    m3mul(R0, R1T, R01);
    m3mul(KiT, R01, A);   // inv(K0)^T R01
    m3mul(A, K1T, B);     // ... K1^T
    m3mul(K1, R1, Cm);    // K1 R1
    m3mul(Cm, R0T, D);    // K1 R1 R0^T
    m3v(R01, T1, w);
    for (int a = 0; a < 3; ++a) u[a] = T0[a] - w[a];
    m3v(D, u, v);
    double S[9] = {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0}, Fd[9];
    m3mul(B, S, Fd);
    float f32[9], sum = 0.f;
    for (int a = 0; a < 9; ++a) { f32[a] = (float)Fd[a]; sum = faddr(sum, f32[a]); }
    for (int a = 0; a < 9; ++a) F[idx * 9 + a] = (sum == 0.f) ? faddr(f32[a], 1e-12f) : f32[a];
}

// ------------------------------------------------------------------------------------------------
// affinity
// ------------------------------------------------------------------------------------------------
// NumPy's float32 pairwise summation (loops_utils.h.src, PW_BLOCKSIZE = 128), same operation order.
__device__ __noinline__ float np_pairwise_leaf_f32(const float* a, int n) {
    MVMC_ASSUME_LDS(a);
    if (n < 8) {
        float r = -0.0f;
        for (int i = 0; i < n; ++i) r = faddr(r, a[i]);
        return r;
    }
    float r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
        r0 = faddr(r0, a[i]); r1 = faddr(r1, a[i + 1]); r2 = faddr(r2, a[i + 2]); r3 = faddr(r3, a[i + 3]);
        r4 = faddr(r4, a[i + 4]); r5 = faddr(r5, a[i + 5]); r6 = faddr(r6, a[i + 6]); r7 = faddr(r7, a[i + 7]);
    }
    float res = faddr(faddr(faddr(r0, r1), faddr(r2, r3)), faddr(faddr(r4, r5), faddr(r6, r7)));
    for (; i < n; ++i) res = faddr(res, a[i]);
    return res;
}
__device__ __noinline__ float np_pairwise_sum_f32(const float* a, int n) {
    MVMC_ASSUME_LDS(a);
    int s_[16], n_[16], st_[16];
    float left_[16];
    int sp = 0;
    s_[0] = 0; n_[0] = n; st_[0] = 0;
    float ret = 0.f;
    while (sp >= 0) {
        int s = s_[sp], m = n_[sp];
        if (st_[sp] == 0) {
            if (m <= 128) { ret = np_pairwise_leaf_f32(a + s, m); --sp; continue; }
            int n2 = m / 2; n2 -= n2 % 8;
            st_[sp] = 1;
            ++sp; s_[sp] = s; n_[sp] = n2; st_[sp] = 0;
        } else if (st_[sp] == 1) {
            int n2 = m / 2; n2 -= n2 % 8;
            left_[sp] = ret; st_[sp] = 2;
            ++sp; s_[sp] = s + n2; n_[sp] = m - n2; st_[sp] = 0;
        } else {
            ret = faddr(left_[sp], ret);
            --sp;
        }
    }
    return ret;
}

// mean_j |line_j . x_j| over 17 joints in NumPy's reduction order (8 accumulators + tail)
__device__ inline double mean17(const double* d) {
    double r[8];
    for (int k = 0; k < 8; ++k) r[k] = dadd(d[k], d[8 + k]);
    double res = dadd(dadd(dadd(r[0], r[1]), dadd(r[2], r[3])), dadd(dadd(r[4], r[5]), dadd(r[6], r[7])));
    res = dadd(res, d[16]);
    return res / 17.0;
}

// distance of pose b's joints to the epipolar lines of pose a's joints:
// line = normalise(F^T [x_a, 1]) (computeCorrespondEpilines(pts, 2, F)); mv_math_util.py:307-315
// (the result goes through *out, a word in the CALLER's frame: only a call that is handed a pointer into its caller's frame is not a
// tail-call candidate, and only then does the compiler drop the callee-saved convention of this local function -- with it every call
// saved and restored the ~50 callee-saved vector registers it touches, 100 scratch stores per call and lane; now 37, and as many in
// the caller.  mvmc_chain.hip has the story.)
__device__ __noinline__ void proj_dist(const double* pa, const double* pb, const float* F, double* out) {
    MVMC_ASSUME_LDS(pa); MVMC_ASSUME_LDS(pb);
    double f[9];
    for (int k = 0; k < 9; ++k) f[k] = (double)F[k];
    double d[17];
    for (int j = 0; j < 17; ++j) {
        double x = pa[j * 3], y = pa[j * 3 + 1];
        double a = dadd(dadd(dmul(f[0], x), dmul(f[3], y)), f[6]);
        double b = dadd(dadd(dmul(f[1], x), dmul(f[4], y)), f[7]);
        double c = dadd(dadd(dmul(f[2], x), dmul(f[5], y)), f[8]);
        double nu = dadd(dmul(a, a), dmul(b, b));
        double sc = (nu != 0.0) ? 1.0 / sqrt(nu) : 1.0;
        a = dmul(a, sc); b = dmul(b, sc); c = dmul(c, sc);
        double v = dadd(dadd(dmul(a, pb[j * 3]), dmul(b, pb[j * 3 + 1])), c);
        d[j] = fabs(v);
    }
    *out = mean17(d);
}

// One frame on the calling wave.  sm: N*51 doubles + 2*N*N floats + 2*N ints + 4 words of LDS.
__device__ __forceinline__ void affinity_wave(double* sm, const double* __restrict__ kps17, const int32_t* __restrict__ counts,
                                              const float* __restrict__ Fm, int C, int P, int f, float* __restrict__ Do,
                                              float* __restrict__ So) {
    const int tid = threadIdx.x & 63;
    const int N = C * P;
    double* pts = sm;                                   // [N][17][3] compact node order
    float* D = reinterpret_cast<float*>(pts + N * 51);  // [n*n] contiguous (ld = n)
    float* tmp = D + N * N;                             // [n*n]
    int* node_q = reinterpret_cast<int*>(tmp + N * N);  // [N] node -> local pose index c*P+p
    int* node_v = node_q + N;                           // [N] node -> view
    int& s_n = node_v[N];
    float& s_mean = reinterpret_cast<float*>(node_v + N)[1];
    float& s_std = reinterpret_cast<float*>(node_v + N)[2];
    if (tid == 0) {
        int n = 0;
        for (int c = 0; c < C; ++c) {
            int cnt = counts[f * C + c];
            cnt = cnt < 0 ? 0 : (cnt > P ? P : cnt);
            for (int p = 0; p < cnt; ++p) { node_q[n] = c * P + p; node_v[n] = c; ++n; }
        }
        s_n = n;
    }
    MVMC_WAVE_SYNC();
    const int n = s_n;
    const double* src = kps17 + (size_t)f * N * 51;
    for (int e = tid; e < n * 51; e += 64) {
        int i = e / 51, r = e - i * 51;
        pts[e] = src[node_q[i] * 51 + r];
    }
    for (int e = tid; e < n * n; e += 64) {
        int i = e / n, j = e - i * n;
        D[e] = (i == j) ? 0.f : 50.f;
    }
    MVMC_WAVE_SYNC();
    // every unordered node pair of different views
    for (int e = tid; e < n * n; e += 64) {
        int i = e / n, j = e - i * n;
        if (j <= i) continue;
        int a = node_v[i], b = node_v[j];
        if (a == b) continue;
        double d_ab, d_ba;
        proj_dist(pts + i * 51, pts + j * 51, Fm + (a * C + b) * 9, &d_ab);
        proj_dist(pts + j * 51, pts + i * 51, Fm + (b * C + a) * 9, &d_ba);
        float v = (float)dmul(0.5, dadd(d_ab, d_ba));
        D[i * n + j] = v;
        D[j * n + i] = v;
    }
    MVMC_WAVE_SYNC();
    // float32 statistics in NumPy's order (mv_math_util.py:348): mean, population std
    const int nn = n * n;
    if (tid == 0 && nn > 0) s_mean = np_pairwise_sum_f32(D, nn) / (float)nn;
    MVMC_WAVE_SYNC();
    for (int e = tid; e < nn; e += 64) {
        float x = D[e] - s_mean;
        tmp[e] = fmulr(x, x);
    }
    MVMC_WAVE_SYNC();
    if (tid == 0 && nn > 0) s_std = sqrtf(np_pairwise_sum_f32(tmp, nn) / (float)nn);
    MVMC_WAVE_SYNC();
    for (int e = tid; e < N * N; e += 64) {
        int i = e / N, j = e - i * N;
        float d = 0.f, s = 0.f;
        if (i < n && j < n) {
            d = D[i * n + j];
            float a = -(d - s_mean) / s_std;
            float t = fmulr(-5.f, a);
            float ex = np_exp_f32(t);          // NumPy's float32 exp, which is not the correctly-rounded one (mvmc_common.h)
            s = __fdiv_rn(1.f, faddr(1.f, ex));
        }
        if (Do) Do[e] = d;
        if (So) So[e] = s;
    }
}

__global__ void __launch_bounds__(64, MVMC_SMALL_WPS)   // (proj_dist is shared with the chain kernel, which runs three workgroups per CU)
affinity_kernel(const double* __restrict__ kps17, const int32_t* __restrict__ counts, const float* __restrict__ Fm,
                int C, int P, float* __restrict__ Dg, float* __restrict__ Sg) {
    extern __shared__ double sm[];
    const int f = blockIdx.x, N = C * P;
    affinity_wave(sm, kps17, counts, Fm, C, P, f, Dg ? Dg + (size_t)f * N * N : nullptr, Sg ? Sg + (size_t)f * N * N : nullptr);
}

// ------------------------------------------------------------------------------------------------
// ALS matching + closure + cluster labels
// ------------------------------------------------------------------------------------------------
template <int NT>
__device__ inline double block_sum(double v, double* red) {
    v = wave_sum(v);
    if constexpr (NT > 64) {
        const int w = threadIdx.x >> 6;
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[w] = v;
        __syncthreads();
        v = 0;
        for (int k = 0; k < NT / 64; ++k) v += red[k];
    }
    return v;
}

// In-place Gauss-Jordan on M (r x ld, ld = r + n): left block SPD (G + ridge), right block RHS.
// After the sweep row a holds diag * solution; no pivoting is needed for an SPD system.
template <int NT>
__device__ inline void gauss_jordan(double* M, int r, int ld) {
    for (int p = 0; p < r; ++p) {
        const double inv = 1.0 / M[p * ld + p];
        const int w = ld - p - 1;
        for (int idx = threadIdx.x; idx < (r - 1) * w; idx += NT) {
            int a = idx / w, b = p + 1 + (idx - a * w);
            if (a >= p) ++a;
            M[a * ld + b] -= M[a * ld + p] * (M[p * ld + b] * inv);
        }
        __syncthreads();
    }
}

// LDS of the generic ALS (any n <= NMAX, rank <= RMAX): X1 as a dense matrix, both factors, the augmented normal matrix
template <int NMAX, int RMAX, int NT>
struct AlsGenLds {
    double sX[NMAX * NMAX];
    double sA[NMAX * RMAX];
    double sB[NMAX * RMAX];
    double sM[RMAX * (RMAX + NMAX)];
    double sRed[NT / 64 + 1];
    int sGid[NMAX];
    uint8_t sVis[NMAX];
    int sKeep[NMAX];
    int s_n, s_r;
};

// One graph (index f of the batch) on an NT-thread workgroup; every thread of the workgroup must call it.  als_kernel is the
// stand-alone wrapper; the chain kernel's large layout (C8 P8) calls it from its persistent workgroup.
template <typename TW, int NMAX, int RMAX, int NT>
__device__ __forceinline__ void als_gen_graph(AlsGenLds<NMAX, RMAX, NT>& L, int f, const TW* __restrict__ W,
                                              const int32_t* __restrict__ gcounts, int G, int ldw,
                                              const double* __restrict__ seed, int seed_len, uint8_t* __restrict__ x_bin,
                                              uint8_t* __restrict__ match_mat, int32_t* __restrict__ labels,
                                              int32_t* __restrict__ n_clusters, int32_t* __restrict__ iters_out) {
    constexpr int T = (NMAX * NMAX + NT - 1) / NT;
    double *sX = L.sX, *sA = L.sA, *sB = L.sB, *sM = L.sM, *sRed = L.sRed;
    int *sGid = L.sGid, *sKeep = L.sKeep;
    uint8_t* sVis = L.sVis;
    int &s_n = L.s_n, &s_r = L.s_r;
    const int tid = threadIdx.x;
    __syncthreads();   // the LDS may still be in use by the caller's previous phase

    if (tid == 0) {
        int n = 0, total = 0, gmax = 0;
        for (int g = 0; g < G; ++g) {
            int c = gcounts[f * G + g];
            c = c < 0 ? 0 : c;
            total += c;
            for (int k = 0; k < c && n < NMAX; ++k) sGid[n++] = g;
            if (c > gmax) gmax = c;
        }
        s_n = total;
        int r = 2 * gmax;
        s_r = r < total ? r : total;
    }
    __syncthreads();
    const int n = s_n, r = s_r;
    int32_t* lab = labels + (size_t)f * ldw;
    if (n == 0 || n > NMAX || n > ldw || r > RMAX || n * r > seed_len) {
        for (int i = tid; i < ldw; i += NT) lab[i] = -1;
        if (tid == 0) { n_clusters[f] = 0; iters_out[f] = (n == 0) ? 0 : -1; }
        return;
    }

    // ---- per-thread element state (registers): W, Z, Y, previous X ----
    int ei[T], ej[T];
    double w[T], z[T], y[T], xp[T];
    float w32[T];
    const TW* Wf = W + (size_t)f * ldw * ldw;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        int e = t * NT + tid;
        bool ok = e < n * n;
        int i = ok ? e / n : 0, j = ok ? e - i * n : 0;
        ei[t] = ok ? i : -1; ej[t] = j;
        if constexpr (sizeof(TW) == 4) {
            float a = ok ? (float)Wf[i * ldw + j] : 0.f, b = ok ? (float)Wf[j * ldw + i] : 0.f;
            w32[t] = fmulr(0.5f, faddr(a, b));
            w[t] = (double)w32[t];
        } else {
            double a = ok ? (double)Wf[i * ldw + j] : 0., b = ok ? (double)Wf[j * ldw + i] : 0.;
            w[t] = 0.5 * (a + b);
            w32[t] = 0.f;
        }
        z[t] = w[t]; xp[t] = w[t]; y[t] = 0.0;
    }
    for (int e = tid; e < n * r; e += NT) sA[e] = seed[e];
    __syncthreads();

    double mu = 64.0;
    const int ld = r + n;
    int iters = 1000;
    for (int it = 0; it < 1000; ++it) {
        // X1 = Z - (Y - W + beta) / mu      (float32 arithmetic on iteration 1 when W is f32)
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (ei[t] < 0) continue;
            double x1;
            if (sizeof(TW) == 4 && it == 0) {
                float q = faddr(-w32[t], 0.1f) / 64.f;
                x1 = (double)(w32[t] - q);
            } else {
                x1 = z[t] - ((y[t] - w[t]) + 0.1) / mu;
            }
            sX[t * NT + tid] = x1;
        }
        __syncthreads();
        const double ridge = 50.0 / mu;
        // M = [A^T A + ridge I | A^T X1]
        for (int idx = tid; idx < r * ld; idx += NT) {
            int a = idx / ld, b = idx - a * ld;
            double s = 0;
            if (b < r) {
                for (int k = 0; k < n; ++k) s += sA[k * r + a] * sA[k * r + b];
                if (a == b) s += ridge;
            } else {
                for (int k = 0; k < n; ++k) s += sA[k * r + a] * sX[k * n + (b - r)];
            }
            sM[idx] = s;
        }
        __syncthreads();
        gauss_jordan<NT>(sM, r, ld);
        for (int idx = tid; idx < n * r; idx += NT) {
            int j = idx / r, a = idx - j * r;
            sB[idx] = sM[a * ld + r + j] / sM[a * ld + a];
        }
        __syncthreads();
        // M = [B^T B + ridge I | B^T X1^T]
        for (int idx = tid; idx < r * ld; idx += NT) {
            int a = idx / ld, b = idx - a * ld;
            double s = 0;
            if (b < r) {
                for (int k = 0; k < n; ++k) s += sB[k * r + a] * sB[k * r + b];
                if (a == b) s += ridge;
            } else {
                for (int k = 0; k < n; ++k) s += sB[k * r + a] * sX[(b - r) * n + k];
            }
            sM[idx] = s;
        }
        __syncthreads();
        gauss_jordan<NT>(sM, r, ld);
        for (int idx = tid; idx < n * r; idx += NT) {
            int j = idx / r, a = idx - j * r;
            sA[idx] = sM[a * ld + r + j] / sM[a * ld + a];
        }
        __syncthreads();
        // X = A B^T ; Z ; Y ; residuals
        double acc_p = 0, acc_d = 0;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (ei[t] < 0) continue;
            const int i = ei[t], j = ej[t];
            double x = 0;
            for (int a = 0; a < r; ++a) x += sA[i * r + a] * sB[j * r + a];
            double zz = x + y[t] / mu;
            if (sGid[i] == sGid[j]) zz = 0.0;
            if (i == j) zz = 1.0;
            zz = zz < 0.0 ? 0.0 : (zz > 1.0 ? 1.0 : zz);
            const double dz = x - zz, dx = x - xp[t];
            y[t] = y[t] + mu * dz;
            z[t] = zz;
            xp[t] = x;
            acc_p += dz * dz;
            acc_d += dx * dx;
        }
        const double p_res = sqrt(block_sum<NT>(acc_p, sRed)) / n;
        const double d_res = mu * sqrt(block_sum<NT>(acc_d, sRed)) / n;
        if (p_res < 1e-4 && d_res < 1e-4) { iters = it + 1; break; }
        if (p_res > 10 * d_res) mu = 2 * mu;
        else if (d_res > 10 * p_res) mu = mu / 2;
    }

    // ---- X_bin = (X + X^T)/2 > 0.5 ----  (sA / sB are free now: reuse them for the byte matrices)
    static_assert(NMAX * NMAX <= NMAX * RMAX * 8, "byte matrices must fit the factor buffers");
    uint8_t* sBin = reinterpret_cast<uint8_t*>(sA);
    uint8_t* sOut = reinterpret_cast<uint8_t*>(sB);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < T; ++t)
        if (ei[t] >= 0) sX[t * NT + tid] = xp[t];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < T; ++t)
        if (ei[t] >= 0) sBin[t * NT + tid] = (0.5 * (sX[ei[t] * n + ej[t]] + sX[ej[t] * n + ei[t]])) > 0.5;
    __syncthreads();
    // ---- transform_closure: only k = n-1 survives (mv_association.py:105-110) ----
    for (int e = tid; e < n * n; e += NT) {
        int i = e / n, j = e - i * n;
        sOut[e] = 0;
        // reuse the upper bits of sBin for temp to save LDS: bit0 = x_bin, bit1 = temp
        uint8_t xb = sBin[e] & 1;
        uint8_t tmpv = xb | ((sBin[i * n + (n - 1)] & 1) & (sBin[(n - 1) * n + j] & 1));
        sX[e] = (double)tmpv;  // temp kept in sX (free now)
    }
    for (int i = tid; i < n; i += NT) sVis[i] = 0;
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        const bool skip = sVis[i] != 0;
        __syncthreads();
        if (!skip)
            for (int j = tid; j < n; j += NT)
                if (sX[i * n + j] != 0.0) { sVis[j] = 1; sOut[j * n + i] = 1; }
        __syncthreads();
    }
    // ---- parse_match_result rule: keep columns with >= 2 members, first kept column wins ----
    for (int c = tid; c < n; c += NT) {
        int s = 0;
        for (int j = 0; j < n; ++j) s += sOut[j * n + c];
        sKeep[c] = s >= 2;
    }
    __syncthreads();
    for (int row = tid; row < ldw; row += NT) {
        int label = -1;
        if (row < n) {
            int ord = 0;
            for (int c = 0; c < n; ++c) {
                if (!sKeep[c]) continue;
                if (sOut[row * n + c]) { label = ord; break; }
                ++ord;
            }
        }
        lab[row] = label;
    }
    if (tid == 0) {
        int k = 0;
        for (int c = 0; c < n; ++c) k += sKeep[c];
        n_clusters[f] = k;
        iters_out[f] = iters;
    }
    if (x_bin || match_mat) {
        for (int e = tid; e < ldw * ldw; e += NT) {
            int i = e / ldw, j = e - i * ldw;
            bool in = i < n && j < n;
            if (x_bin) x_bin[(size_t)f * ldw * ldw + e] = in ? (sBin[i * n + j] & 1) : 0;
            if (match_mat) match_mat[(size_t)f * ldw * ldw + e] = in ? sOut[i * n + j] : 0;
        }
    }
}

template <typename TW, int NMAX, int RMAX, int NT>
__global__ void __launch_bounds__(NT)
als_kernel(const TW* __restrict__ W, const int32_t* __restrict__ gcounts, int G, int ldw,
           const double* __restrict__ seed, int seed_len, uint8_t* __restrict__ x_bin,
           uint8_t* __restrict__ match_mat, int32_t* __restrict__ labels, int32_t* __restrict__ n_clusters,
           int32_t* __restrict__ iters_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char als_gen_lds[];
    auto& L = *reinterpret_cast<AlsGenLds<NMAX, RMAX, NT>*>(als_gen_lds);
    als_gen_graph<TW, NMAX, RMAX, NT>(L, blockIdx.x, W, gcounts, G, ldw, seed, seed_len, x_bin, match_mat, labels, n_clusters, iters_out);
}

// ------------------------------------------------------------------------------------------------
// ALS, latency-oriented variant for n <= 32 nodes (one wave per frame).
// Lane (i = lane & 31, h = lane >> 5) owns row i, column half h of W/Z/Y/X in registers.  Both factor
// updates are "R x R normal matrix + one right-hand side per lane": the normal matrix is eliminated once
// by lanes that hold its rows in registers (pivot rows broadcast with shuffles, no LDS round trips), and
// every lane then applies the stored multipliers to its own right-hand side.  All inner loops have
// compile-time bounds (nodes are padded to NMAX with exact zeros, which changes no sum).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = r * (2.0 - x * r);
    r = r * (2.0 - x * r);
    return r;
}

// Gauss-Jordan elimination of the R x R matrix whose row (lane % R) sits in g[]; writes the multipliers
// mul[p][a] (row a, pivot p) and the reciprocal pivots dinv[a].
template <int R>
__device__ __forceinline__ void gj_chain(double (&g)[R], double* __restrict__ mul, double* __restrict__ dinv) {
    const int lane = threadIdx.x, a = lane % R;
#pragma unroll
    for (int p = 0; p < R; ++p) {
        double piv[R];
#pragma unroll
        for (int b = p; b < R; ++b) {
            // broadcast row p (lane p) through SGPRs: v_readlane, no LDS crossbar round trip
            const unsigned long long bits = __double_as_longlong(g[b]);
            const unsigned lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffu), p);
            const unsigned hi = __builtin_amdgcn_readlane((int)(bits >> 32), p);
            piv[b] = __longlong_as_double(((unsigned long long)hi << 32) | lo);
        }
        const double m = (a == p) ? 0.0 : g[p] * fast_rcp(piv[p]);
#pragma unroll
        for (int b = p + 1; b < R; ++b) g[b] -= m * piv[b];
        if (lane < R) mul[p * R + a] = m;
    }
    // g[a] with a per-lane a: a chain of selects over static indices (a dynamic index would send the whole array through
    // scratch memory -- eight stores and a dependent load from global memory on the critical path of every half-iteration)
    double ga = g[0];
#pragma unroll
    for (int b = 1; b < R; ++b) ga = (a == b) ? g[b] : ga;
    if (lane < R) dinv[a] = 1.0 / ga;
}

template <int R>
__device__ __forceinline__ void gj_apply(double (&hv)[R], const double* __restrict__ mul, const double* __restrict__ dinv) {
#pragma unroll
    for (int p = 0; p < R; ++p) {
        const double hp = hv[p];
        const double2* mr = reinterpret_cast<const double2*>(&mul[p * R]);
#pragma unroll
        for (int a = 0; a < R; a += 2) { const double2 v2 = mr[a >> 1]; hv[a] -= v2.x * hp; hv[a + 1] -= v2.y * hp; }  // mul[p][p] == 0
        hv[p] = hp;
    }
#pragma unroll
    for (int a = 0; a < R; ++a) hv[a] *= dinv[a];
}

#ifdef MVMC_ALS_PROFILE
__shared__ long long g_alsprof[8];
#define APROF(k) { const long long _t = clock64(); if (threadIdx.x == 0) g_alsprof[k] += _t - _tp; _tp = _t; }
// als7: work / wait per phase as seen by the solver wave (thread 0, slots 0-13) and a worker wave (thread 64, slots 14-27)
__shared__ long long g_alsprof2[28];
#define APROF2(k) { const long long _t = clock64(); if ((threadIdx.x & 0xbf) == 0) g_alsprof2[(threadIdx.x >> 6) * 14 + (k)] += _t - _tp; _tp = _t; }
#else
#define APROF(k)
#define APROF2(k)
#endif
template <typename TW, int NMAX, int R>
__device__ __forceinline__ int als2_iterate(const TW* __restrict__ Wf, int ldw, int n, int r, const int* sGid,
                                            const double* __restrict__ seed, double* sX, double* sA, double* sB,
                                            double* sG, double* sMul, double* sDinv) {
    constexpr int NH = NMAX / 2, LDX = NMAX + 1;
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    const bool row_ok = i < n;
    double w[NH], z[NH], y[NH], xp[NH], x1[NH];
    float w32[NH];
    unsigned valid = 0, same = 0;
#pragma unroll
    for (int c = 0; c < NH; ++c) {
        const int j = h * NH + c;
        const bool ok = row_ok && j < n;
        if (ok) valid |= 1u << c;
        if (ok && sGid[i] == sGid[j]) same |= 1u << c;
        if constexpr (sizeof(TW) == 4) {
            const float a = ok ? (float)Wf[i * ldw + j] : 0.f, b = ok ? (float)Wf[j * ldw + i] : 0.f;
            w32[c] = fmulr(0.5f, faddr(a, b));
            w[c] = (double)w32[c];
        } else {
            const double a = ok ? (double)Wf[i * ldw + j] : 0., b = ok ? (double)Wf[j * ldw + i] : 0.;
            w[c] = 0.5 * (a + b);
            w32[c] = 0.f;
        }
        z[c] = w[c]; xp[c] = w[c]; y[c] = 0.0;
    }
    // factor A (rows in LDS, read by every lane)
    for (int e = lane; e < NMAX * R; e += 64) {
        const int k = e / R, a = e - k * R;
        sA[e] = (k < n && a < r) ? seed[k * r + a] : 0.0;
    }
    __syncthreads();
    double mu = 64.0;
    int iters = 1000;
    const int n4 = (n + 3) & ~3;  // rows beyond n are exact zeros: loops stop at the next multiple of 4
#ifdef MVMC_ALS_PROFILE
    if (threadIdx.x < 8) g_alsprof[threadIdx.x] = 0;
    long long _tp = clock64();
#endif
    for (int it = 0; it < 1000; ++it) {
        // ---- X1 = Z - (Y - W + beta)/mu ; own row half in registers, whole matrix in LDS ----
        const double inv_mu = 1.0 / mu;  // mu = 64 * 2^k: the reciprocal is exact, x * inv_mu == x / mu bit for bit
#pragma unroll
        for (int c = 0; c < NH; ++c) {
            double v = 0.0;
            if ((valid >> c) & 1) {
                if (sizeof(TW) == 4 && it == 0) v = (double)(w32[c] - faddr(-w32[c], 0.1f) / 64.f);
                else v = z[c] - ((y[c] - w[c]) + 0.1) * inv_mu;
            }
            x1[c] = v;
            if (i < NMAX) sX[i * LDX + h * NH + c] = v;
        }
        __syncthreads();
        APROF(0)
        const double ridge = 50.0 / mu;
        double hv[R];
        // ---- B update: G = A^T A + ridge I ; H[:, j] = A^T X1[:, j] ----
        for (int e = lane; e < R * R; e += 64) {
            const int a = e / R, b = e - a * R;
            double g0 = (a == b) ? ridge : 0.0, g1 = 0.0;
#pragma unroll 4
            for (int k = 0; k < n4; k += 2) { g0 += sA[k * R + a] * sA[k * R + b]; g1 += sA[(k + 1) * R + a] * sA[(k + 1) * R + b]; }
            sG[e] = g0 + g1;
        }
#pragma unroll
        for (int a = 0; a < R; ++a) hv[a] = 0.0;
        if (i < NMAX) {
            const int k_end = (h + 1) * NH < n4 ? (h + 1) * NH : n4;
#pragma unroll 4
            for (int k = h * NH; k < k_end; ++k) {
                const double xv = sX[k * LDX + i];
                const double2* ar = reinterpret_cast<const double2*>(&sA[k * R]);
#pragma unroll
                for (int a = 0; a < R; a += 2) { const double2 v2 = ar[a >> 1]; hv[a] += v2.x * xv; hv[a + 1] += v2.y * xv; }
            }
        }
#pragma unroll
        for (int a = 0; a < R; ++a) hv[a] += __shfl_xor(hv[a], 32, 64);
        __syncthreads();
        APROF(1)
        {
            double g[R];
#pragma unroll
            for (int b = 0; b < R; ++b) g[b] = sG[(lane % R) * R + b];
            gj_chain<R>(g, sMul, sDinv);
        }
        __syncthreads();
        APROF(2)
        gj_apply<R>(hv, sMul, sDinv);  // hv = B[i][:]
        if (h == 0 && i < NMAX)
#pragma unroll
            for (int a = 0; a < R; ++a) sB[i * R + a] = hv[a];
        __syncthreads();
        APROF(3)
        // ---- A update: G = B^T B + ridge I ; H[:, i] = B^T X1[i, :]^T (own row, registers) ----
        for (int e = lane; e < R * R; e += 64) {
            const int a = e / R, b = e - a * R;
            double g0 = (a == b) ? ridge : 0.0, g1 = 0.0;
#pragma unroll 4
            for (int k = 0; k < n4; k += 2) { g0 += sB[k * R + a] * sB[k * R + b]; g1 += sB[(k + 1) * R + a] * sB[(k + 1) * R + b]; }
            sG[e] = g0 + g1;
        }
        double av[R];
#pragma unroll
        for (int a = 0; a < R; ++a) av[a] = 0.0;
#pragma unroll
        for (int c = 0; c < NH; ++c) {
            const double xv = x1[c];
            const double2* br = reinterpret_cast<const double2*>(&sB[(h * NH + c) * R]);
#pragma unroll
            for (int a = 0; a < R; a += 2) { const double2 v2 = br[a >> 1]; av[a] += v2.x * xv; av[a + 1] += v2.y * xv; }
        }
#pragma unroll
        for (int a = 0; a < R; ++a) av[a] += __shfl_xor(av[a], 32, 64);
        __syncthreads();
        APROF(4)
        {
            double g[R];
#pragma unroll
            for (int b = 0; b < R; ++b) g[b] = sG[(lane % R) * R + b];
            gj_chain<R>(g, sMul, sDinv);
        }
        __syncthreads();
        APROF(5)
        gj_apply<R>(av, sMul, sDinv);  // av = A[i][:]
        if (h == 0 && i < NMAX)
#pragma unroll
            for (int a = 0; a < R; ++a) sA[i * R + a] = av[a];
        // ---- X = A B^T, Z, Y, residuals ----
        double acc_p = 0.0, acc_d = 0.0;
#pragma unroll
        for (int c = 0; c < NH; ++c) {
            const int j = h * NH + c;
            double xa = 0.0, xb = 0.0;
            const double2* br = reinterpret_cast<const double2*>(&sB[j * R]);
#pragma unroll
            for (int a = 0; a < R; a += 2) { const double2 v2 = br[a >> 1]; xa += av[a] * v2.x; xb += av[a + 1] * v2.y; }
            const double x = xa + xb;
            if ((valid >> c) & 1) {
                double zz = x + y[c] * inv_mu;
                if ((same >> c) & 1) zz = 0.0;
                if (i == j) zz = 1.0;
                zz = zz < 0.0 ? 0.0 : (zz > 1.0 ? 1.0 : zz);
                const double dz = x - zz, dx = x - xp[c];
                y[c] = y[c] + mu * dz;
                z[c] = zz;
                xp[c] = x;
                acc_p += dz * dz;
                acc_d += dx * dx;
            }
        }
        const double p_res = sqrt(wave_sum(acc_p)) / n;
        const double d_res = mu * sqrt(wave_sum(acc_d)) / n;
        __syncthreads();  // sA complete before the next iteration reads it
        APROF(6)
        if (p_res < 1e-4 && d_res < 1e-4) { iters = it + 1; break; }
        if (p_res > 10 * d_res) mu = 2 * mu;
        else if (d_res > 10 * p_res) mu = mu / 2;
    }
    // final X (dense n x n, leading dimension n) for the symmetrise / binarise tail
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NH; ++c)
        if ((valid >> c) & 1) sX[i * n + h * NH + c] = xp[c];
    __syncthreads();
    return iters;
}

template <typename TW, int NMAX>
__global__ void __launch_bounds__(64)
als2_kernel(const TW* __restrict__ W, const int32_t* __restrict__ gcounts, int G, int ldw,
            const double* __restrict__ seed, int seed_len, uint8_t* __restrict__ x_bin,
            uint8_t* __restrict__ match_mat, int32_t* __restrict__ labels, int32_t* __restrict__ n_clusters,
            int32_t* __restrict__ iters_out) {
    constexpr int RMAX = 16;
    __shared__ double sX[NMAX * (NMAX + 1)];
    __shared__ __attribute__((aligned(16))) double sA[NMAX * RMAX];
    __shared__ __attribute__((aligned(16))) double sB[NMAX * RMAX];
    __shared__ double sG[RMAX * RMAX];
    __shared__ __attribute__((aligned(16))) double sMul[RMAX * RMAX];
    __shared__ double sDinv[RMAX];
    __shared__ int sGid[NMAX];
    __shared__ uint8_t sVis[NMAX];
    __shared__ int sKeep[NMAX];
    __shared__ int s_n, s_r;
    const int f = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        int n = 0, total = 0, gmax = 0;
        for (int g = 0; g < G; ++g) {
            int c = gcounts[f * G + g];
            c = c < 0 ? 0 : c;
            total += c;
            for (int k = 0; k < c && n < NMAX; ++k) sGid[n++] = g;
            if (c > gmax) gmax = c;
        }
        s_n = total;
        const int r = 2 * gmax;
        s_r = r < total ? r : total;
    }
    __syncthreads();
    const int n = s_n, r = s_r;
    int32_t* lab = labels + (size_t)f * ldw;
    if (n == 0 || n > NMAX || n > ldw || r > RMAX || n * r > seed_len) {
        for (int i = tid; i < ldw; i += 64) lab[i] = -1;
        if (tid == 0) { n_clusters[f] = 0; iters_out[f] = (n == 0) ? 0 : -1; }
        return;
    }
    const TW* Wf = W + (size_t)f * ldw * ldw;
    const int iters = (r <= 8) ? als2_iterate<TW, NMAX, 8>(Wf, ldw, n, r, sGid, seed, sX, sA, sB, sG, sMul, sDinv)
                               : als2_iterate<TW, NMAX, 16>(Wf, ldw, n, r, sGid, seed, sX, sA, sB, sG, sMul, sDinv);
    // ---- tail: X_bin, closure (k = n-1 only), labels -- same rules as als_kernel ----
    uint8_t* sBin = reinterpret_cast<uint8_t*>(sA);
    uint8_t* sOut = reinterpret_cast<uint8_t*>(sB);
    uint8_t* sTmp = reinterpret_cast<uint8_t*>(sG);
    static_assert(NMAX * NMAX <= NMAX * RMAX * 8 && NMAX * NMAX <= RMAX * RMAX * 8, "byte matrices must fit");
    for (int e = tid; e < n * n; e += 64) {
        const int i = e / n, j = e - i * n;
        sBin[e] = (0.5 * (sX[i * n + j] + sX[j * n + i])) > 0.5;
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += 64) {
        const int i = e / n, j = e - i * n;
        sOut[e] = 0;
        sTmp[e] = sBin[e] | (sBin[i * n + (n - 1)] & sBin[(n - 1) * n + j]);
    }
    for (int i = tid; i < n; i += 64) sVis[i] = 0;
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        const bool skip = sVis[i] != 0;
        __syncthreads();
        if (!skip)
            for (int j = tid; j < n; j += 64)
                if (sTmp[i * n + j]) { sVis[j] = 1; sOut[j * n + i] = 1; }
        __syncthreads();
    }
    for (int c = tid; c < n; c += 64) {
        int s = 0;
        for (int j = 0; j < n; ++j) s += sOut[j * n + c];
        sKeep[c] = s >= 2;
    }
    __syncthreads();
    for (int row = tid; row < ldw; row += 64) {
        int label = -1;
        if (row < n) {
            int ord = 0;
            for (int c = 0; c < n; ++c) {
                if (!sKeep[c]) continue;
                if (sOut[row * n + c]) { label = ord; break; }
                ++ord;
            }
        }
        lab[row] = label;
    }
    if (tid == 0) {
        int k = 0;
        for (int c = 0; c < n; ++c) k += sKeep[c];
        n_clusters[f] = k;
        iters_out[f] = iters;
#ifdef MVMC_ALS_PROFILE
        for (int q = 0; q < 7; ++q) lab[ldw - 7 + q] = (int)(g_alsprof[q] / iters);
#endif
    }
    if (x_bin || match_mat) {
        for (int e = tid; e < ldw * ldw; e += 64) {
            const int i = e / ldw, j = e - i * ldw;
            const bool in = i < n && j < n;
            if (x_bin) x_bin[(size_t)f * ldw * ldw + e] = in ? sBin[i * n + j] : 0;
            if (match_mat) match_mat[(size_t)f * ldw * ldw + e] = in ? sOut[i * n + j] : 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ALS, latency-oriented variant for few frames per launch: one 256-thread workgroup per frame.
// The temporal path advances all chains one frame per launch, so a launch holds only a few hundred graphs and
// its duration is the iteration count of the slowest graph times the latency of one iteration; als2 (one wave
// per graph) leaves three quarters of each CU idle there.  Here thread (i = tid & 31, h = tid >> 5) owns row i
// and the column slice [h NS, (h+1) NS) of W/Z/Y/X (NS = NMAX / 8), so the element-wise work, the A^T X1 / B^T X1
// accumulations and X = A B^T are spread over four waves; the slices' partial sums meet in LDS.  The R x R normal
// matrices are formed and eliminated by wave 0 (gj_chain) while the other waves accumulate, and one lane per
// row applies the multipliers.  Same arithmetic per element as als2 except for the summation order of the
// partial sums.
// ------------------------------------------------------------------------------------------------
template <typename TW, int NMAX, int R>
__device__ __forceinline__ int als4_iterate(const TW* __restrict__ Wf, int ldw, int n, int r, const int* sGid,
                                            const double* __restrict__ seed, double* sX, double* sA, double* sB,
                                            double* sG, double* sMul, double* sDinv, double* sHp, double* sRed) {
    constexpr int NS = NMAX / 8, LDX = NMAX + 1;
    const int tid = threadIdx.x, i = tid & 31, h = tid >> 5, wv = tid >> 6, lane = tid & 63;
    const bool row_ok = i < n && i < NMAX;
    double w[NS], z[NS], y[NS], xp[NS], x1[NS];
    float w32[NS];
    unsigned valid = 0, same = 0;
#pragma unroll
    for (int c = 0; c < NS; ++c) {
        const int j = h * NS + c;
        const bool ok = row_ok && j < n;
        if (ok) valid |= 1u << c;
        if (ok && sGid[i] == sGid[j]) same |= 1u << c;
        if constexpr (sizeof(TW) == 4) {
            const float a = ok ? (float)Wf[i * ldw + j] : 0.f, b = ok ? (float)Wf[j * ldw + i] : 0.f;
            w32[c] = fmulr(0.5f, faddr(a, b));
            w[c] = (double)w32[c];
        } else {
            const double a = ok ? (double)Wf[i * ldw + j] : 0., b = ok ? (double)Wf[j * ldw + i] : 0.;
            w[c] = 0.5 * (a + b);
            w32[c] = 0.f;
        }
        z[c] = w[c]; xp[c] = w[c]; y[c] = 0.0;
    }
    for (int e = tid; e < NMAX * R; e += 256) {
        const int k = e / R, a = e - k * R;
        sA[e] = (k < n && a < r) ? seed[k * r + a] : 0.0;
    }
    __syncthreads();
    double mu = 64.0, inv_mu = 1.0 / 64.0;  // mu = 64 * 2^k: the reciprocal is exact, x * inv_mu == x / mu bit for bit
    int iters = 1000;
#ifdef MVMC_ALS_PROFILE
    if (threadIdx.x < 8) g_alsprof[threadIdx.x] = 0;
    long long _tp = clock64();
#endif
    const int n4 = (n + 3) & ~3;  // rows beyond n are exact zeros: loops stop at the next multiple of 4
    // normal matrix of factor F (rows in LDS): R = 16 has one entry per thread; R = 8 has 64 entries, so each wave sums a
    // quarter of the rows for all of them (sG[wave][entry]) and the eliminating wave adds the four parts
    constexpr bool SPLITG = (R == 8);
    auto normal_matrix = [&](const double* F, double ridge) {
        if constexpr (SPLITG) {
            const int e = tid & 63, a = e / R, b = e - a * R;
            const int q4 = n4 >> 2, k0 = wv * q4;
            double g0 = (wv == 0 && a == b) ? ridge : 0.0;
            for (int k = k0; k < k0 + q4; ++k) g0 += F[k * R + a] * F[k * R + b];
            sG[wv * 64 + e] = g0;
        } else if (tid < R * R) {
            const int a = tid / R, b = tid - a * R;
            double g0 = (a == b) ? ridge : 0.0, g1 = 0.0;
#pragma unroll 4
            for (int k = 0; k < n4; k += 2) { g0 += F[k * R + a] * F[k * R + b]; g1 += F[(k + 1) * R + a] * F[(k + 1) * R + b]; }
            sG[tid] = g0 + g1;
        }
    };
    // slices' partial right-hand sides -> sHp[wave][row][R]; the two slices of a wave are added by shuffle
    auto publish_partial = [&](double (&hv)[R]) {
#pragma unroll
        for (int a = 0; a < R; ++a) hv[a] += __shfl_xor(hv[a], 32, 64);
        if ((lane & 32) == 0 && i < NMAX) {
            double2* dst = reinterpret_cast<double2*>(&sHp[(wv * NMAX + i) * R]);
#pragma unroll
            for (int a = 0; a < R; a += 2) dst[a >> 1] = make_double2(hv[a], hv[a + 1]);
        }
    };
    // wave 0 eliminates sG; then one lane per row sums the four partial right-hand sides, applies the
    // multipliers and writes the factor row
    auto solve_rows = [&](double* Fout) {
        __syncthreads();                       // sG and sHp complete
        APROF(1)
        if (wv == 0) {
            double g[R];
#pragma unroll
            for (int b = 0; b < R; ++b) {
                const int e = (lane % R) * R + b;
                if constexpr (SPLITG) g[b] = (sG[e] + sG[64 + e]) + (sG[128 + e] + sG[192 + e]);
                else g[b] = sG[e];
            }
            gj_chain<R>(g, sMul, sDinv);
        }
        __syncthreads();
        APROF(2)
        if (tid < NMAX) {
            double hv[R];
#pragma unroll
            for (int a = 0; a < R; a += 2) {
                const double2 p0 = *reinterpret_cast<const double2*>(&sHp[(0 * NMAX + tid) * R + a]);
                const double2 p1 = *reinterpret_cast<const double2*>(&sHp[(1 * NMAX + tid) * R + a]);
                const double2 p2 = *reinterpret_cast<const double2*>(&sHp[(2 * NMAX + tid) * R + a]);
                const double2 p3 = *reinterpret_cast<const double2*>(&sHp[(3 * NMAX + tid) * R + a]);
                hv[a] = (p0.x + p1.x) + (p2.x + p3.x);
                hv[a + 1] = (p0.y + p1.y) + (p2.y + p3.y);
            }
            gj_apply<R>(hv, sMul, sDinv);
            double2* dst = reinterpret_cast<double2*>(&Fout[tid * R]);
#pragma unroll
            for (int a = 0; a < R; a += 2) dst[a >> 1] = make_double2(hv[a], hv[a + 1]);
        }
        __syncthreads();
        APROF(3)
    };
    for (int it = 0; it < 1000; ++it) {
        // ---- X1 = Z - (Y - W + beta)/mu ----
#pragma unroll
        for (int c = 0; c < NS; ++c) {
            double v = 0.0;
            if ((valid >> c) & 1) {
                if (sizeof(TW) == 4 && it == 0) v = (double)(w32[c] - faddr(-w32[c], 0.1f) / 64.f);
                else v = z[c] - ((y[c] - w[c]) + 0.1) * inv_mu;
            }
            x1[c] = v;
            if (i < NMAX) sX[i * LDX + h * NS + c] = v;
        }
        __syncthreads();
        APROF(0)
        const double ridge = 50.0 * inv_mu;  // == 50 / mu exactly
        // ---- B update: (A^T A + ridge I) B[i]^T = A^T X1[:, i]; slice h covers rows k of its column range ----
        normal_matrix(sA, ridge);
        {
            double hv[R];
#pragma unroll
            for (int a = 0; a < R; ++a) hv[a] = 0.0;
            if (i < NMAX) {
#pragma unroll
                for (int c = 0; c < NS; ++c) {
                    const int k = h * NS + c;
                    const double xv = sX[k * LDX + i];
                    const double2* ar = reinterpret_cast<const double2*>(&sA[k * R]);
#pragma unroll
                    for (int a = 0; a < R; a += 2) { const double2 v2 = ar[a >> 1]; hv[a] += v2.x * xv; hv[a + 1] += v2.y * xv; }
                }
            }
            publish_partial(hv);
        }
        solve_rows(sB);
        // ---- A update: (B^T B + ridge I) A[i]^T = B^T X1[i, :]^T (own row, own columns) ----
        normal_matrix(sB, ridge);
        {
            double av[R];
#pragma unroll
            for (int a = 0; a < R; ++a) av[a] = 0.0;
#pragma unroll
            for (int c = 0; c < NS; ++c) {
                const double xv = x1[c];
                const double2* br = reinterpret_cast<const double2*>(&sB[(h * NS + c) * R]);
#pragma unroll
                for (int a = 0; a < R; a += 2) { const double2 v2 = br[a >> 1]; av[a] += v2.x * xv; av[a + 1] += v2.y * xv; }
            }
            publish_partial(av);
        }
        solve_rows(sA);
        // ---- X = A B^T, Z, Y, residuals ----
        double av[R];
        {
            const double2* ar = reinterpret_cast<const double2*>(&sA[(i < NMAX ? i : 0) * R]);
#pragma unroll
            for (int a = 0; a < R; a += 2) { const double2 v2 = ar[a >> 1]; av[a] = v2.x; av[a + 1] = v2.y; }
        }
        double acc_p = 0.0, acc_d = 0.0;
#pragma unroll
        for (int c = 0; c < NS; ++c) {
            const int j = h * NS + c;
            double xa = 0.0, xb = 0.0;
            const double2* br = reinterpret_cast<const double2*>(&sB[j * R]);
#pragma unroll
            for (int a = 0; a < R; a += 2) { const double2 v2 = br[a >> 1]; xa += av[a] * v2.x; xb += av[a + 1] * v2.y; }
            const double x = xa + xb;
            if ((valid >> c) & 1) {
                double zz = x + y[c] * inv_mu;
                if ((same >> c) & 1) zz = 0.0;
                if (i == j) zz = 1.0;
                zz = zz < 0.0 ? 0.0 : (zz > 1.0 ? 1.0 : zz);
                const double dz = x - zz, dx = x - xp[c];
                y[c] = y[c] + mu * dz;
                z[c] = zz;
                xp[c] = x;
                acc_p += dz * dz;
                acc_d += dx * dx;
            }
        }
        APROF(4)
        acc_p = wave_sum_dpp(acc_p); acc_d = wave_sum_dpp(acc_d);
        if (lane == 0) { sRed[wv] = acc_p; sRed[4 + wv] = acc_d; }
        __syncthreads();
        APROF(5)
        const double p_res = sqrt((sRed[0] + sRed[1]) + (sRed[2] + sRed[3])) / n;
        const double d_res = mu * sqrt((sRed[4] + sRed[5]) + (sRed[6] + sRed[7])) / n;
        __syncthreads();  // sRed is rewritten by the next iteration
        APROF(6)
        if (p_res < 1e-4 && d_res < 1e-4) { iters = it + 1; break; }
        if (p_res > 10 * d_res) { mu = 2 * mu; inv_mu = 0.5 * inv_mu; }
        else if (d_res > 10 * p_res) { mu = mu / 2; inv_mu = 2 * inv_mu; }
    }
    // final X (dense n x n, leading dimension n) for the symmetrise / binarise tail
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NS; ++c)
        if ((valid >> c) & 1) sX[i * n + h * NS + c] = xp[c];
    __syncthreads();
    return iters;
}

// ------------------------------------------------------------------------------------------------
// als7: solver wave + row-group workers (rank <= 8, n <= 24).  als4 spends a half-iteration as accumulate (all waves) | barrier |
// eliminate (wave 0) | barrier | apply (one lane per row) | barrier.  Here:
//  * workers own a row with EIGHT lanes (three columns each; waves 1-3 hold rows 0-7, 8-15, 16-23), so a row's right-hand side is
//    complete inside its wave after three DPP steps -- no partials through LDS, no per-row serial back-substitution;
//  * the solver wave turns the normal matrix into its explicit inverse (Gauss-Jordan on [G | I], one row of both halves per lane, the
//    pivot row through v_readlane); the workers apply it as one 8-term dot product per lane (lane c of a row group makes entry c);
//  * the inverse of A^T A + ridge I for the B update is made while the workers do the X/Z/Y update and the X1 step (A^T A is formed
//    before the residuals are known, the ridge -- which depends on mu -- is added after), so only the inverse for the A update
//    remains on the critical path.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int src) {
    const unsigned long long bits = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffu), src);
    const unsigned hi = __builtin_amdgcn_readlane((int)(bits >> 32), src);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
// Gauss-Jordan steps P0 .. P1-1 on [G | E]: lane (c, a) = (lane >> 3, lane & 7) holds row a of G in registers (eight replicas) and entry
// (a, c) of E.  The pivot row of G travels through SGPRs (v_readlane), the pivot row of E through one ds_bpermute.  (Tried: all of E in
// registers too, rows through v_readlane only -- 16 readlanes per step instead of 9 on average, 13% slower.)
// (Also tried, round 2: ONE entry of G and of E per lane, the pivot row of both through the LDS crossbar, the row's own pivot-column entry by
// DPP row_newbcast -- the form that pays for the 16 x 16 inversion of als5.  Bit-identical, but here every pivot then waits for a crossbar
// round trip beside three worker waves that keep the LDS busy: pivots 3-7 went 0.91 k -> 1.15 k cycles, ALS 22.4 -> 24.2 M cycles per chain.)
// (Round 3: the pivots in 2 x 2 blocks, as in als5 -- one reciprocal and one crossbar round trip per two pivots.  Exact against the
// oracle's iteration counts, but slower here: a block needs both pivot rows in SGPRs, 4 (8 - p) v_readlane instead of 2 (8 - p) +
// 2 (7 - p), and the chain it shortens is not what a pivot of this layout waits for: form + inversion 2.27 k -> 2.35 k cycles,
// 490 k -> 480 k frames/s on the same box.  Also tried: B^T B as three partial sums made by the worker waves over the rows they
// have just written -- the solver's form 0.77 k -> 0.30 k, the workers' apply phase + 0.35 k: 491 k -> 489 k.)
template <int P0, int P1>
__device__ __forceinline__ void gj_inv_steps(double (&g)[8], double& e, double& dself, int lane) {
    const int a = lane & 7;
#pragma unroll
    for (int p = P0; p < P1; ++p) {
        double piv[8];
#pragma unroll
        for (int b = p; b < 8; ++b) piv[b] = readlane_f64(g[b], p);   // lane p: row p
        const double ep = __shfl(e, (lane & ~7) | p, 64);             // row p of E, same column
        const double rinv = fast_rcp(piv[p]);
        dself = (a == p) ? rinv : dself;
        const double m = (a == p) ? 0.0 : g[p] * rinv;
#pragma unroll
        for (int b = p + 1; b < 8; ++b) g[b] -= m * piv[b];
        e -= m * ep;
    }
}
// sum over the eight lanes of a row group (lanes 8q .. 8q+7); every lane gets the result
__device__ __forceinline__ double oct_sum(double v) {
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror: the other quad of the group
    return v;
}

template <typename TW, int NMAX>
__device__ __forceinline__ int als7_iterate(const TW* __restrict__ Wf, int ldw, int n, int r, const int* sGid,
                                            const double* __restrict__ seed, double* sX, double* sA, double* sB,
                                            double* sG, double* sInvA, double* sInvB, double* sRed) {
    constexpr int R = 8, NS = 3, LDX = NMAX + 1;
    // Rows of the factors, of the Gram matrix and of the inverses are LDF = 10 doubles apart, not 8: a worker's sixteen-byte reads of row
    // k = 3 sub + c (eight different rows per quarter-wave) then start 60 banks apart instead of 48 and no two of them share a bank --
    // with 64-byte rows, rows k and k + 12 (sub and sub + 4) did, and every such read took two passes (a third of the ALS's LDS cycles
    // were bank conflicts: SQ_LDS_BANK_CONFLICT)
    constexpr int LDF = 10;
    static_assert(NMAX * LDF <= NMAX * 16 && 2 * R * LDF <= 4 * NMAX * 16 && R * LDF <= 256, "padded rows fit the arrays of Als4Lds");
    static_assert(NMAX >= 24, "three worker waves of eight rows");
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const bool worker = wv > 0;
    const int sub = lane & 7;                                  // worker: column slice / output entry; solver: row a
    const int i = worker ? (wv - 1) * 8 + (lane >> 3) : 0;     // worker: row 0 .. 23
    const bool row_ok = worker && i < n;
    double w[NS], z[NS], y[NS], xp[NS], x1[NS];
    float w32[NS];
    unsigned valid = 0, same = 0;
#pragma unroll
    for (int c = 0; c < NS; ++c) {
        const int j = sub * NS + c;
        const bool ok = row_ok && j < n;
        if (ok) valid |= 1u << c;
        if (ok && sGid[i] == sGid[j]) same |= 1u << c;
        if constexpr (sizeof(TW) == 4) {
            const float a = ok ? (float)Wf[i * ldw + j] : 0.f, b = ok ? (float)Wf[j * ldw + i] : 0.f;
            w32[c] = fmulr(0.5f, faddr(a, b));
            w[c] = (double)w32[c];
        } else {
            const double a = ok ? (double)Wf[i * ldw + j] : 0., b = ok ? (double)Wf[j * ldw + i] : 0.;
            w[c] = 0.5 * (a + b);
            w32[c] = 0.f;
        }
        z[c] = w[c]; xp[c] = w[c]; y[c] = 0.0; x1[c] = 0.0;
    }
    for (int e = tid; e < NMAX * R; e += 256) {
        const int k = e / R, a = e - k * R;
        sA[k * LDF + a] = (k < n && a < r) ? seed[k * r + a] : 0.0;
        sB[k * LDF + a] = 0.0;
    }
    __syncthreads();
    double mu = 64.0, inv_mu = 1.0 / 64.0;  // mu = 64 * 2^k: the reciprocal is exact
    int iters = 1000;
#ifdef MVMC_ALS_PROFILE
    if (threadIdx.x < 28) g_alsprof2[threadIdx.x] = 0;
    __syncthreads();
    long long _tp = clock64();
#endif
    const double tol2 = (1e-4 * n) * (1e-4 * n);
    // ---- solver wave ----
    double g0[R];   // row (lane & 7) of F^T F
    // (tried: two entries per lane over half of the rows each, fully unrolled: 3% more cycles per iteration in the chain kernel)
    // (tried: the three rows of B a worker reads for the A update's right-hand side kept in registers for the X = A B^T step, which reads the
    // same rows -- 48 more live registers: 6 -> 34 scratch stores in the phase, ALS 23.5 -> 29.8 M cycles per chain)
    auto form = [&](const double* F) {
        const int a = lane >> 3, b = lane & 7;   // one entry per lane, two accumulators over even / odd rows
        double s0 = 0.0, s1 = 0.0;
        // (all 24 rows, fully unrolled: the rows beyond n are exact zeros and add nothing, and with a fixed trip count every load of the
        // sum is in flight at once -- behind a run-time bound the loop paid an LDS round trip per four rows, and for B^T B the solver
        // wave's form + inversion is what the workers wait for)
        static_assert(NMAX >= 24, "rows of the factors");
#pragma unroll
        for (int k = 0; k < 24; k += 2) { s0 += F[k * LDF + a] * F[k * LDF + b]; s1 += F[(k + 1) * LDF + a] * F[(k + 1) * LDF + b]; }
        sG[a * LDF + b] = s0 + s1;
        MVMC_WAVE_SYNC();
        const double2* gr = reinterpret_cast<const double2*>(&sG[(lane & 7) * LDF]);
#pragma unroll
        for (int c = 0; c < R; c += 2) { const double2 v2 = gr[c >> 1]; g0[c] = v2.x; g0[c + 1] = v2.y; }
        MVMC_WAVE_SYNC();
    };
    double g[R], ge = 0.0, gd = 0.0;
    auto inv_begin = [&](double ridge) {
        const int a = lane & 7, c = lane >> 3;
#pragma unroll
        for (int b = 0; b < R; ++b) g[b] = g0[b] + ((a == b) ? ridge : 0.0);
        ge = (a == c) ? 1.0 : 0.0;
        gd = 0.0;
    };
    auto inv_store = [&](double* dst) { dst[(lane & 7) * LDF + (lane >> 3)] = ge * gd; };   // E[a][c] / pivot a
    // ---- workers ----
    auto rhs = [&](const double* F, bool transposed, double (&hv)[R]) {   // sum_k F[k] xs(k, i) over the group's 24 columns
#pragma unroll
        for (int a = 0; a < R; ++a) hv[a] = 0.0;
#pragma unroll
        for (int c = 0; c < NS; ++c) {
            const int k = sub * NS + c;
            const double xv = transposed ? sX[k * LDX + i] : x1[c];
            const double2* fr = reinterpret_cast<const double2*>(&F[k * LDF]);
#pragma unroll
            for (int a = 0; a < R; a += 2) { const double2 v2 = fr[a >> 1]; hv[a] += v2.x * xv; hv[a + 1] += v2.y * xv; }
        }
#pragma unroll
        for (int a = 0; a < R; ++a) hv[a] = oct_sum(hv[a]);
    };
    auto apply = [&](const double* inv, const double (&hv)[R], double* Fout) {   // entry `sub` of inv . hv
        const double2* gr = reinterpret_cast<const double2*>(&inv[sub * LDF]);
        double o0 = 0.0, o1 = 0.0;
#pragma unroll
        for (int a = 0; a < R; a += 2) { const double2 v2 = gr[a >> 1]; o0 += v2.x * hv[a]; o1 += v2.y * hv[a + 1]; }
        Fout[i * LDF + sub] = o0 + o1;
    };
#if MVMC_SMALL_WPS >= 4
    // ---- the 128-register form (a fourth workgroup per CU: -DMVMC_SMALL_WPS=4; measured in round 5) ----
    // The end of an iteration, the same on every wave: residuals p_res = sqrt(sP) / n, d_res = mu sqrt(sD) / n only feed thresholds
    // (stop; mu x 2 or / 2): decide them on the squared sums (mu is a power of two, so mu^2 sD is exact), and only when a comparison is
    // closer than 1e-9 to its threshold take the IEEE sqrt / divide path that NumPy's expressions round through (uniform over the
    // workgroup).  Returns true when the iteration stops.
    auto decide = [&]() {
        const double sP = (sRed[0] + sRed[1]) + sRed[2], sD = (mu * mu) * ((sRed[3] + sRed[4]) + sRed[5]);
        bool stop = sP < tol2 && sD < tol2, up = sP > 100.0 * sD, down = sD > 100.0 * sP;
        {
            const double eps = 1e-9;
            const bool amb = !(sP > 1e-200) || !(sD > 1e-200) || fabs(sP - tol2) <= eps * tol2 || fabs(sD - tol2) <= eps * tol2 ||
                             fabs(sP - 100.0 * sD) <= eps * sP || fabs(sD - 100.0 * sP) <= eps * sD;
            if (amb) {
                asm volatile("" ::: "memory");   // (keeps this a branch: the IEEE sqrt / divide sequences stay off the common path)
                const double p_res = sqrt(sP) / n, d_res = mu * sqrt((sRed[3] + sRed[4]) + sRed[5]) / n;
                stop = p_res < 1e-4 && d_res < 1e-4; up = p_res > 10 * d_res; down = d_res > 10 * p_res;
            }
        }
        // (sRed is next written after five more barriers)
        if (stop) return true;
        if (up) { mu = 2 * mu; inv_mu = 0.5 * inv_mu; }
        else if (down) { mu = mu / 2; inv_mu = 2 * inv_mu; }
        return false;
    };
    // The solver wave and the worker waves run SEPARATE loops with the same six barriers per iteration (a barrier counts arrivals, not
    // addresses): each role gets its own register allocation -- in one loop the allocator keeps the union live (the solver's two copies
    // of the Gram matrix beside the workers' W / Z / Y / X state and right-hand sides), which a 128-register build can only hold with
    // a dozen scratch reloads per iteration.
    if (!worker) {
#ifdef MVMC_PRIO_SOLVER
        __builtin_amdgcn_s_setprio(MVMC_PRIO_SOLVER);
#endif
        // The first three pivots of (A^T A + ridge I)^-1 are made AHEAD, in the solver wave's idle time at the end of the previous iteration
        // (it waits ~600 cycles for the workers' X / Z / Y step), with the ridge of that iteration: mu changes in few iterations, and only
        // then are they made again here -- the workers' X1 step used to wait for them (solver 1.0 k cycles against 0.5 k).
        double ridge_made = 50.0 * inv_mu;
        form(sA); inv_begin(ridge_made); gj_inv_steps<0, 3>(g, ge, gd, lane);
        for (int it = 0; it < 1000; ++it) {
            const double ridge = 50.0 * inv_mu;  // == 50 / mu exactly
            if (ridge != ridge_made) {   // (wave-uniform: mu moved at the end of the last iteration)
                inv_begin(ridge);
                gj_inv_steps<0, 3>(g, ge, gd, lane);
            }
            APROF2(0)
            __syncthreads();
            APROF2(1)
            gj_inv_steps<3, 8>(g, ge, gd, lane); inv_store(sInvA);
            APROF2(2)
            __syncthreads();
            APROF2(3)
            APROF2(4)
            __syncthreads();
            APROF2(5)
            form(sB); inv_begin(ridge); gj_inv_steps<0, 8>(g, ge, gd, lane); inv_store(sInvB);
            APROF2(6)
            __syncthreads();
            APROF2(7)
            APROF2(8)
            __syncthreads();
            APROF2(9)
            // A^T A for the next iteration
            form(sA);
            ridge_made = ridge;   // speculation: mu stays
            inv_begin(ridge_made);
            gj_inv_steps<0, 3>(g, ge, gd, lane);
            APROF2(10)
            __syncthreads();
            APROF2(11)
            const bool stop = decide();
            APROF2(12)
            if (stop) { iters = it + 1; break; }
        }
#ifdef MVMC_PRIO_SOLVER
        __builtin_amdgcn_s_setprio(0);
#endif
    } else {
        for (int it = 0; it < 1000; ++it) {
            double hv[R];
            // ---- X1 = Z - (Y - W + beta)/mu (own entries; the matrix also goes to LDS for the transposed reads) ----
#pragma unroll
            for (int c = 0; c < NS; ++c) {
                double v = 0.0;
                if ((valid >> c) & 1) {
                    if (sizeof(TW) == 4 && it == 0) v = (double)(w32[c] - faddr(-w32[c], 0.1f) / 64.f);
                    else v = z[c] - ((y[c] - w[c]) + 0.1) * inv_mu;
                }
                x1[c] = v;
                sX[i * LDX + sub * NS + c] = v;
            }
            APROF2(0)
            __syncthreads();
            APROF2(1)
            // ---- B update: right-hand sides A^T X1[:, i] (column i of X1: through LDS) ----
            rhs(sA, true, hv);
            APROF2(2)
            __syncthreads();
            APROF2(3)
            apply(sInvA, hv, sB);
            APROF2(4)
            __syncthreads();
            APROF2(5)
            // ---- A update: right-hand sides B^T X1[i, :]^T (own row, own columns) ----
            rhs(sB, false, hv);
            APROF2(6)
            __syncthreads();
            APROF2(7)
            apply(sInvB, hv, sA);
            APROF2(8)
            __syncthreads();
            APROF2(9)
            // ---- X = A B^T, Z, Y, residuals ----
            double acc_p = 0.0, acc_d = 0.0;
            {
                double av[R];
                {
                    const double2* ar = reinterpret_cast<const double2*>(&sA[i * LDF]);
#pragma unroll
                    for (int a = 0; a < R; a += 2) { const double2 v2 = ar[a >> 1]; av[a] = v2.x; av[a + 1] = v2.y; }
                }
#pragma unroll
                for (int c = 0; c < NS; ++c) {
                    const int j = sub * NS + c;
                    double xa = 0.0, xb = 0.0;
                    const double2* br = reinterpret_cast<const double2*>(&sB[j * LDF]);
#pragma unroll
                    for (int a = 0; a < R; a += 2) { const double2 v2 = br[a >> 1]; xa += av[a] * v2.x; xb += av[a + 1] * v2.y; }
                    const double x = xa + xb;
                    if ((valid >> c) & 1) {
                        double zz = x + y[c] * inv_mu;
                        if ((same >> c) & 1) zz = 0.0;
                        if (i == j) zz = 1.0;
                        zz = zz < 0.0 ? 0.0 : (zz > 1.0 ? 1.0 : zz);
                        const double dz = x - zz, dx = x - xp[c];
                        y[c] = y[c] + mu * dz;
                        z[c] = zz;
                        xp[c] = x;
                        acc_p += dz * dz;
                        acc_d += dx * dx;
                    }
                }
                acc_p = wave_sum_dpp(acc_p); acc_d = wave_sum_dpp(acc_d);
                if (lane == 0) { sRed[wv - 1] = acc_p; sRed[3 + wv - 1] = acc_d; }
            }
            APROF2(10)
            __syncthreads();
            APROF2(11)
            const bool stop = decide();
            APROF2(12)
            if (stop) { iters = it + 1; break; }
        }
    }
#else
    // The first three pivots of (A^T A + ridge I)^-1 are made AHEAD, in the solver wave's idle time at the end of the previous iteration
    // (it waits ~600 cycles for the workers' X / Z / Y step), with the ridge of that iteration: mu changes in few iterations, and only
    // then are they made again here -- the workers' X1 step used to wait for them (solver 1.0 k cycles against 0.5 k).
    double ridge_made = 50.0 * inv_mu;
    if (!worker) { form(sA); inv_begin(ridge_made); gj_inv_steps<0, 3>(g, ge, gd, lane); }
    for (int it = 0; it < 1000; ++it) {
        const double ridge = 50.0 * inv_mu;  // == 50 / mu exactly
        double hv[R];
        // ---- X1 = Z - (Y - W + beta)/mu (own entries; the matrix also goes to LDS for the transposed reads) ----
        if (worker) {
#pragma unroll
            for (int c = 0; c < NS; ++c) {
                double v = 0.0;
                if ((valid >> c) & 1) {
                    if (sizeof(TW) == 4 && it == 0) v = (double)(w32[c] - faddr(-w32[c], 0.1f) / 64.f);
                    else v = z[c] - ((y[c] - w[c]) + 0.1) * inv_mu;
                }
                x1[c] = v;
                sX[i * LDX + sub * NS + c] = v;
            }
        } else if (ridge != ridge_made) {   // (wave-uniform: mu moved at the end of the last iteration)
            inv_begin(ridge);
            gj_inv_steps<0, 3>(g, ge, gd, lane);
        }
        APROF2(0)
        __syncthreads();
        APROF2(1)
        // ---- B update: right-hand sides A^T X1[:, i] (column i of X1: through LDS) ----
        if (worker) rhs(sA, true, hv);
        else { gj_inv_steps<3, 8>(g, ge, gd, lane); inv_store(sInvA); }
        APROF2(2)
        __syncthreads();
        APROF2(3)
        if (worker) apply(sInvA, hv, sB);
        APROF2(4)
        __syncthreads();
        APROF2(5)
        // ---- A update: right-hand sides B^T X1[i, :]^T (own row, own columns) ----
        if (worker) rhs(sB, false, hv);
        else { form(sB); inv_begin(ridge); gj_inv_steps<0, 8>(g, ge, gd, lane); inv_store(sInvB); }
        APROF2(6)
        __syncthreads();
        APROF2(7)
        if (worker) apply(sInvB, hv, sA);
        APROF2(8)
        __syncthreads();
        APROF2(9)
        // ---- X = A B^T, Z, Y, residuals (workers); A^T A for the next iteration (solver) ----
        double acc_p = 0.0, acc_d = 0.0;
        if (worker) {
            double av[R];
            {
                const double2* ar = reinterpret_cast<const double2*>(&sA[i * LDF]);
#pragma unroll
                for (int a = 0; a < R; a += 2) { const double2 v2 = ar[a >> 1]; av[a] = v2.x; av[a + 1] = v2.y; }
            }
#pragma unroll
            for (int c = 0; c < NS; ++c) {
                const int j = sub * NS + c;
                double xa = 0.0, xb = 0.0;
                const double2* br = reinterpret_cast<const double2*>(&sB[j * LDF]);
#pragma unroll
                for (int a = 0; a < R; a += 2) { const double2 v2 = br[a >> 1]; xa += av[a] * v2.x; xb += av[a + 1] * v2.y; }
                const double x = xa + xb;
                if ((valid >> c) & 1) {
                    double zz = x + y[c] * inv_mu;
                    if ((same >> c) & 1) zz = 0.0;
                    if (i == j) zz = 1.0;
                    zz = zz < 0.0 ? 0.0 : (zz > 1.0 ? 1.0 : zz);
                    const double dz = x - zz, dx = x - xp[c];
                    y[c] = y[c] + mu * dz;
                    z[c] = zz;
                    xp[c] = x;
                    acc_p += dz * dz;
                    acc_d += dx * dx;
                }
            }
            acc_p = wave_sum_dpp(acc_p); acc_d = wave_sum_dpp(acc_d);
            if (lane == 0) { sRed[wv - 1] = acc_p; sRed[3 + wv - 1] = acc_d; }
        } else {
            form(sA);
            ridge_made = ridge;   // speculation: mu stays
            inv_begin(ridge_made);
            gj_inv_steps<0, 3>(g, ge, gd, lane);
        }
        APROF2(10)
        __syncthreads();
        APROF2(11)
        // residuals p_res = sqrt(sP) / n, d_res = mu sqrt(sD) / n only feed thresholds (stop; mu x 2 or / 2): decide them on the squared
        // sums (mu is a power of two, so mu^2 sD is exact), and only when a comparison is closer than 1e-9 to its threshold take the IEEE
        // sqrt / divide path that NumPy's expressions round through (uniform over the workgroup)
        const double sP = (sRed[0] + sRed[1]) + sRed[2], sD = (mu * mu) * ((sRed[3] + sRed[4]) + sRed[5]);
        bool stop = sP < tol2 && sD < tol2, up = sP > 100.0 * sD, down = sD > 100.0 * sP;
        {
            const double eps = 1e-9;
            const bool amb = !(sP > 1e-200) || !(sD > 1e-200) || fabs(sP - tol2) <= eps * tol2 || fabs(sD - tol2) <= eps * tol2 ||
                             fabs(sP - 100.0 * sD) <= eps * sP || fabs(sD - 100.0 * sP) <= eps * sD;
            if (amb) {
                asm volatile("" ::: "memory");   // (keeps this a branch: the IEEE sqrt / divide sequences stay off the common path)
                const double p_res = sqrt(sP) / n, d_res = mu * sqrt((sRed[3] + sRed[4]) + sRed[5]) / n;
                stop = p_res < 1e-4 && d_res < 1e-4; up = p_res > 10 * d_res; down = d_res > 10 * p_res;
            }
        }
        // (sRed is next written after five more barriers)
        APROF2(12)
        if (stop) { iters = it + 1; break; }
        if (up) { mu = 2 * mu; inv_mu = 0.5 * inv_mu; }
        else if (down) { mu = mu / 2; inv_mu = 2 * inv_mu; }
    }
#endif
    // final X (dense n x n, leading dimension n) for the symmetrise / binarise tail
    __syncthreads();
    if (worker) {
#pragma unroll
        for (int c = 0; c < NS; ++c)
            if ((valid >> c) & 1) sX[i * n + sub * NS + c] = xp[c];
    }
    __syncthreads();
    return iters;
}

template <int NMAX>
struct Als4Lds {
    static constexpr int RMAX = 16;
    double sX[NMAX * (NMAX + 1)];
    __attribute__((aligned(16))) double sA[NMAX * RMAX];
    __attribute__((aligned(16))) double sB[NMAX * RMAX];
    __attribute__((aligned(16))) double sHp[4 * NMAX * RMAX];
    double sG[RMAX * RMAX];
    __attribute__((aligned(16))) double sMul[RMAX * RMAX];
    double sDinv[RMAX], sRed[8];
    int sGid[NMAX];
    uint8_t sVis[NMAX];
    int sKeep[NMAX];
    int s_n, s_r;
};

// One graph (index f of the batch) on a 256-thread workgroup; every thread of the workgroup must call it.
template <typename TW, int NMAX>
__device__ __forceinline__ void als4_graph(Als4Lds<NMAX>& L, int f, const TW* __restrict__ W,
                                           const int32_t* __restrict__ gcounts, int G, int ldw,
                                           const double* __restrict__ seed, int seed_len, uint8_t* __restrict__ x_bin,
                                           uint8_t* __restrict__ match_mat, int32_t* __restrict__ labels,
                                           int32_t* __restrict__ n_clusters, int32_t* __restrict__ iters_out) {
    constexpr int RMAX = 16, NT4 = 256;
    double *sX = L.sX, *sA = L.sA, *sB = L.sB, *sHp = L.sHp, *sG = L.sG, *sMul = L.sMul, *sDinv = L.sDinv, *sRed = L.sRed;
    int *sGid = L.sGid, *sKeep = L.sKeep;
    uint8_t* sVis = L.sVis;
    int &s_n = L.s_n, &s_r = L.s_r;
    const int tid = threadIdx.x;
    __syncthreads();   // the arena may still be in use by the caller's previous phase
    if (tid == 0) {
        int n = 0, total = 0, gmax = 0;
        for (int g = 0; g < G; ++g) {
            int c = gcounts[f * G + g];
            c = c < 0 ? 0 : c;
            total += c;
            for (int k = 0; k < c && n < NMAX; ++k) sGid[n++] = g;
            if (c > gmax) gmax = c;
        }
        s_n = total;
        const int r = 2 * gmax;
        s_r = r < total ? r : total;
    }
    __syncthreads();
    const int n = s_n, r = s_r;
    int32_t* lab = labels + (size_t)f * ldw;
    if (n == 0 || n > NMAX || n > ldw || r > RMAX || n * r > seed_len) {
        for (int i = tid; i < ldw; i += NT4) lab[i] = -1;
        if (tid == 0) { n_clusters[f] = 0; iters_out[f] = (n == 0) ? 0 : -1; }
        return;
    }
    const TW* Wf = W + (size_t)f * ldw * ldw;
    const int iters = (r <= 8 && n <= 24) ? als7_iterate<TW, NMAX>(Wf, ldw, n, r, sGid, seed, sX, sA, sB, sG, sHp, sHp + 80, sRed)
                      : (r <= 8)          ? als4_iterate<TW, NMAX, 8>(Wf, ldw, n, r, sGid, seed, sX, sA, sB, sG, sMul, sDinv, sHp, sRed)
                                          : als4_iterate<TW, NMAX, 16>(Wf, ldw, n, r, sGid, seed, sX, sA, sB, sG, sMul, sDinv, sHp, sRed);
    // ---- tail: X_bin, closure (k = n-1 only), labels -- same rules as als_kernel ----
    uint8_t* sBin = reinterpret_cast<uint8_t*>(sA);
    uint8_t* sOut = reinterpret_cast<uint8_t*>(sB);
    uint8_t* sTmp = reinterpret_cast<uint8_t*>(sHp);
    for (int e = tid; e < n * n; e += NT4) {
        const int i = e / n, j = e - i * n;
        sBin[e] = (0.5 * (sX[i * n + j] + sX[j * n + i])) > 0.5;
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += NT4) {
        const int i = e / n, j = e - i * n;
        sOut[e] = 0;
        sTmp[e] = sBin[e] | (sBin[i * n + (n - 1)] & sBin[(n - 1) * n + j]);
    }
    for (int i = tid; i < n; i += NT4) sVis[i] = 0;
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        const bool skip = sVis[i] != 0;
        __syncthreads();
        if (!skip)
            for (int j = tid; j < n; j += NT4)
                if (sTmp[i * n + j]) { sVis[j] = 1; sOut[j * n + i] = 1; }
        __syncthreads();
    }
    for (int c = tid; c < n; c += NT4) {
        int s = 0;
        for (int j = 0; j < n; ++j) s += sOut[j * n + c];
        sKeep[c] = s >= 2;
    }
    __syncthreads();
    for (int row = tid; row < ldw; row += NT4) {
        int label = -1;
        if (row < n) {
            int ord = 0;
            for (int c = 0; c < n; ++c) {
                if (!sKeep[c]) continue;
                if (sOut[row * n + c]) { label = ord; break; }
                ++ord;
            }
        }
        lab[row] = label;
    }
    if (tid == 0) {
        int k = 0;
        for (int c = 0; c < n; ++c) k += sKeep[c];
        n_clusters[f] = k;
        iters_out[f] = iters;
#ifdef MVMC_ALS_PROFILE
        // diagnostic build: cycles per iteration by phase {X1, accumulate G+H, eliminate, apply, X/Z/Y, reduce, residuals}
        if (r <= 8 && n <= 24 && ldw >= 28) { for (int q = 0; q < 28; ++q) lab[q] = (int)(g_alsprof2[q] / iters); }
        else for (int q = 0; q < 7; ++q) lab[ldw - 7 + q] = (int)(g_alsprof[q] / iters);
#endif
    }
    if (x_bin || match_mat) {
        for (int e = tid; e < ldw * ldw; e += NT4) {
            const int i = e / ldw, j = e - i * ldw;
            const bool in = i < n && j < n;
            if (x_bin) x_bin[(size_t)f * ldw * ldw + e] = in ? sBin[i * n + j] : 0;
            if (match_mat) match_mat[(size_t)f * ldw * ldw + e] = in ? sOut[i * n + j] : 0;
        }
    }
}

#ifndef ALS4_WG_PER_CU
#define ALS4_WG_PER_CU 3   // (168 registers, as in the chain kernel: 768 graphs resident instead of 512 at the 198 the compiler takes when left alone)
#endif
template <typename TW, int NMAX>
__global__ void __launch_bounds__(256, ALS4_WG_PER_CU)
als4_kernel(const TW* __restrict__ W, const int32_t* __restrict__ gcounts, int G, int ldw,
            const double* __restrict__ seed, int seed_len, uint8_t* __restrict__ x_bin,
            uint8_t* __restrict__ match_mat, int32_t* __restrict__ labels, int32_t* __restrict__ n_clusters,
            int32_t* __restrict__ iters_out) {
    __shared__ Als4Lds<NMAX> L;
    als4_graph<TW, NMAX>(L, blockIdx.x, W, gcounts, G, ldw, seed, seed_len, x_bin, match_mat, labels, n_clusters, iters_out);
}

// ------------------------------------------------------------------------------------------------
// ALS for the large graphs of config 5 (C8 P8: n <= 72 nodes, rank <= 16) on one 512-thread workgroup
// (match_als, mv_association.py:263-312; same iteration as als_kernel / als4_graph, restated for this size).
//   * W, Z, Y and the previous X live in REGISTERS: thread (ti, tj) of a 24 x 18 grid owns the 3 x 4 tile of elements
//     rows 3 ti .., columns 4 tj .. (432 of the 512 threads; eight waves per CU: two per SIMD, so one wave's LDS round trips and
//     dependent fp64 chains are covered by the other's work);
//   * X1 = Z - (Y - W + beta) / mu is a dense n x n matrix in LDS (row stride NMAX), read by both factor updates;
//   * a factor update  B = (inv(A^T A + rho I) (A^T X1))^T : waves 0..4 form the sixteen-column blocks of H = A^T X1 on the MATRIX
//     CORES (v_mfma_f64_16x16x4_f64: eighteen instructions per block at n = 72, operands read as one batch) while the last wave inverts
//     the 16 x 16 normal matrix by Gauss-Jordan in registers (the reference forms explicit inverses too: np.linalg.inv); then
//     inv(G) H, four matrix instructions per block; the Gram matrices A^T A, B^T B are matrix-core products on the solver wave;
//   * X = A B^T on the 3 x 4 tiles (A rows in registers, one 128-byte B row per column, the next column's row prefetched), the
//     Z / Y update and the residual sums in the same pass.
// Seven workgroup barriers per iteration; no state in scratch memory.  Round 3 moved the dense products onto the matrix cores
// (29 k -> 21.8 k cycles per iteration); an fp64 MFMA keeps the pipe for ~64 cycles, so they pay where whole tiles are full.
// (Tried: the fifth column block -- half empty at n = 72 -- split over the k range on the waves 4 .. 6 so that no SIMD carries two
// full chains: 93.0 k -> 90.7 k frames/s; the solver wave's pivots are the critical path, and more matrix work beside them slows them.)
// ------------------------------------------------------------------------------------------------
// WLDS = false: the symmetrised affinity in a per-graph GLOBAL buffer (NMAX * LD doubles, L2-resident: one 12-double tile read per thread
// and iteration) instead of LDS -- 42.6 KB less.  Same values, same arithmetic.  It was what let an association workgroup and a solver
// workgroup share a CU in round 5's two-kernel form of the BIG chain kernel (+ 7 - 10 %, retired in round 6: DESIGN.md section 9); no
// kernel instantiates it now.
template <int NMAX, bool WLDS = true>
struct Als5Lds {
    // Row strides chosen against LDS bank conflicts (64 banks of 4 bytes): the matrices are walked by rows 3 or 4 apart (the tiles),
    // and a stride of 72 or 16 doubles would put every such row on the same banks (an 18-way conflict on the factor rows)
    static constexpr int LD = NMAX + 2;     // X1 / W / H rows: 74 doubles
    static constexpr int FS = 18;           // factor rows: 16 rank slots + 2
    __attribute__((aligned(16))) double sX[NMAX * LD];
    __attribute__((aligned(16))) double sW[WLDS ? NMAX * LD : 2];      // the symmetrised affinity (constant over the iteration): LDS, not registers
    __attribute__((aligned(16))) double sA[NMAX * FS];
    __attribute__((aligned(16))) double sB[NMAX * FS];
    __attribute__((aligned(16))) double sH[16 * LD];
    __attribute__((aligned(16))) double sG[16 * 34];   // inverse of the normal matrix in columns 16..31 (rows padded to 34 doubles)
    __attribute__((aligned(16))) double sGin[16 * 16]; // the normal matrix on its way to the solver wave
    __attribute__((aligned(16))) double sGp[3][16 * 16];   // B^T B in three partial sums over k (waves 0 .. 2), summed by the solver wave's load
    double sRed[16];
    int sGid[NMAX];
    uint8_t sVis[NMAX];
    int sKeep[NMAX];
    int s_n, s_r;
};

// G = F^T F (+ rho on the diagonal; identity on the unused rank slots) over the rows of the factor F (NMAX x FSG in LDS) into sGin
// (16 x 16), on the MATRIX CORES, one wave: a 16 x 16 x n product, eighteen v_mfma_f64_16x16x4_f64 at n = 72.  add_rho = false leaves
// the raw Gram matrix (rho is not known yet).  (The FMA forms it replaces -- one entry per thread on 256 threads, 2 x 2 blocks on the
// solver wave -- took 3 - 6 k cycles on the critical path.)
// Operand layout (checked on gfx950, tools/mfma_layout_test.hip): lane l gives A[i = l % 16][k = l / 16] and B[k = l / 16][j = l % 16] and
// receives D[i = l / 16 + 4 v][j = l % 16] in its four accumulator registers -- for F^T F both operands are the SAME register,
// F[k0 + l / 16][l % 16].  ~0.6 k cycles where the FMA form above took 3 - 6 k on the solver wave; the sums run over k in the matrix
// core's order, so the result differs from the FMA form's in the last bits (round 3: als5 no longer bit-identical with round 2's; the
// gates are the oracle's X_bin / labels / iteration counts, tests/test_gpu_config5_c8p8.py).
typedef double als5_d4 __attribute__((ext_vector_type(4)));
#ifndef MVMC_ALS5_PRE
#define MVMC_ALS5_PRE 4    // pivots of the A-side inversion made ahead (see the solver wave's loop)
#endif
// rows [k_lo, k_hi) of F only (raw partial sum: add_rho = false) when several waves share the product
template <int FSG>
__device__ __forceinline__ void als5_gram_mfma(const double* __restrict__ sF, int n4, int r, double rho, bool add_rho, double* __restrict__ sGin,
                                               int k_lo = 0) {
    const int li = threadIdx.x & 15, lq = (threadIdx.x & 63) >> 4;
    als5_d4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = k_lo; k0 < n4; k0 += 4) {
        const double f = sF[(k0 + lq) * FSG + li];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(f, f, acc, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int a = lq + 4 * v;
        double g = acc[v];
        if (add_rho) {
            if (a == li) g += rho;
            if (a >= r || li >= r) g = a == li ? 1.0 : 0.0;
        }
        sGin[a * 16 + li] = g;
    }
}

// the solver wave: G -> inv(G) by Gauss-Jordan IN PLACE, IN REGISTERS (G is symmetric positive definite: no pivoting; the reference
// forms the explicit inverse too, np.linalg.inv).  Lane (c, g) = (lane & 15, lane >> 4) holds rows 4 g .. 4 g + 3 of column c in
// cur[4]: a row group is one 16-lane DPP row.  Per pivot p (unrolled: every register index is a constant): the pivot row's entry of
// my column comes from lane (c, p / 4) through the LDS crossbar (one ds_bpermute pair), the pivot itself by v_readlane, and the
// pivot column's entries of my four rows from lane p of MY OWN DPP row by row_newbcast -- VALU moves, no crossbar round trip.
// The arithmetic is that of the elimination on the augmented matrix [G | I] (which the first version of this routine carried in
// eight registers per lane, sixteen crossbar broadcasts per pivot): column 16 + p of the identity part is still e_p when pivot p
// comes, so its new entries are 1 / piv and 0 - M[i][p] * (1 / piv), and they take the place of column p, which becomes e_p and is
// never read again.  Same divisions, same FMAs, bit for bit; 16 dependent pivots of ~300 cycles instead of ~600.
// rho_fix: add rho to the diagonal and put the identity on the unused rank slots while loading (the Gram matrix was stored raw).
template <int SRC>
__device__ __forceinline__ double als5_row_newbcast(double v) {   // lane SRC of the caller's 16-lane row, to the whole row
    int lo = __double2loint(v), hi = __double2hiint(v);
    // (bound_ctrl: every lane is written, so no "old" value has to be materialised -- a v_mov 0 per DPP move otherwise, a sixth of a pivot's instructions)
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + SRC, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + SRC, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// the pivot column's entries of the caller's four rows (p is a compile-time constant after unrolling: the switch folds)
__device__ __forceinline__ void als5_pivot_column(const double (&cur)[4], double (&col)[4], int p) {
#define MVMC_A5C(Z) case Z: _Pragma("unroll") for (int q = 0; q < 4; ++q) col[q] = als5_row_newbcast<Z>(cur[q]); break;
    switch (p) {
        MVMC_A5C(0) MVMC_A5C(1) MVMC_A5C(2) MVMC_A5C(3) MVMC_A5C(4) MVMC_A5C(5) MVMC_A5C(6) MVMC_A5C(7)
        MVMC_A5C(8) MVMC_A5C(9) MVMC_A5C(10) MVMC_A5C(11) MVMC_A5C(12) MVMC_A5C(13) MVMC_A5C(14) MVMC_A5C(15)
        default: break;
    }
#undef MVMC_A5C
}
// parts = 3: the matrix arrives as three partial sums 256 doubles apart (sGp)
__device__ __forceinline__ void als5_inv_load(double (&cur)[4], const double* __restrict__ sGin, int r, double rho, bool rho_fix, int parts = 1) {
    const int lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = g * 4 + q;
        double v = sGin[row * 16 + c];
        if (parts == 3) v = (v + sGin[256 + row * 16 + c]) + sGin[512 + row * 16 + c];
        if (rho_fix) {
            if (row == c) v += rho;
            if (row >= r || c >= r) v = row == c ? 1.0 : 0.0;
        }
        cur[q] = v;
    }
}
// Pivots come in 2 x 2 BLOCKS (round 3): rows p, p + 1 (p even) live in the same lanes (registers p & 3 and (p & 3) + 1 of row group
// p / 4), so one crossbar round trip delivers both pivot rows, ONE reciprocal (of the block's determinant) replaces two, and the update
// is two fused multiply-adds per register -- eight dependent steps per inversion instead of sixteen.  With K = {p, p + 1}, P = M[K][K]:
//   M'[K][j] = inv(P) M[K][j],  M'[i][j] = M[i][j] - M[i][K] inv(P) M[K][j],  and in the columns of K (the identity part of the
//   augmented matrix, as in the scalar form): M'[K][K] = inv(P), M'[i][K] = -M[i][K] inv(P).
// The same elimination as two scalar pivots in exact arithmetic ((a r1 - b' r0) / (a d - b b') is what the second scalar pivot forms
// from the first one's results); the rounding differs.  G is positive definite, so det P > 0.
// (Tried: LOOK-AHEAD -- every lane carries the current block's pivot rows of its column and applies each block's update to the NEXT
// block's rows itself (fetched through the crossbar in their state before the block, their pivot-column entries through v_readlane), so
// that the crossbar round trip leaves the dependent chain.  Exact, bit-identical rows -- and slower: eight more v_readlane and their
// SGPR hazards per block cost more than the crossbar wait they hide; a block 590 -> 680 cycles, 107.8 k -> 104.7 k frames/s.)
template <int P0, int P1>
__device__ __forceinline__ void als5_inv_pivots(double (&cur)[4]) {
    static_assert(P0 % 2 == 0 && P1 % 2 == 0, "pivot blocks are pairs");
    const int lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int p = P0; p < P1; p += 2) {
        const int pg = p >> 2, pq = p & 3;
        const double m0 = cur[pq], m1 = cur[pq + 1];
        double r0 = __shfl(m0, pg * 16 + c, 64), r1 = __shfl(m1, pg * 16 + c, 64);      // M[p][c], M[p + 1][c]
#define MVMC_A5RL(V, L) __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(V), (L)), __builtin_amdgcn_readlane(__double2loint(V), (L)))
        const double a = MVMC_A5RL(m0, pg * 16 + p), b = MVMC_A5RL(m0, pg * 16 + p + 1);      // M[p][p], M[p][p + 1]
        const double bt = MVMC_A5RL(m1, pg * 16 + p), d = MVMC_A5RL(m1, pg * 16 + p + 1);     // M[p + 1][p], M[p + 1][p + 1]
#undef MVMC_A5RL
        double c0[4], c1[4];
        als5_pivot_column(cur, c0, p);                              // M[4 g + q][p]
        als5_pivot_column(cur, c1, p + 1);                          // M[4 g + q][p + 1]
        const bool in_k = c == p || c == p + 1;
        if (in_k) { r0 = c == p ? 1.0 : 0.0; r1 = c == p ? 0.0 : 1.0; }   // (columns 16 + p, 16 + p + 1 of the augmented matrix)
        // (tried on the scalar form: the pivot row through v_permlane swaps instead of the LDS crossbar -- 77.9 k -> 76.9 k frames/s; the
        // pivot's reciprocal taken out of an IEEE division bit-exactly -- slower; v_rcp_f64 + two Newton steps, ~1 ulp, instead of the
        // division -- kept: the division's ~25 dependent instructions were half of a pivot's 380 cycles.)
        const double idet = fast_rcp64(fma(a, d, -(b * bt)));
        const double f0 = (d * r0 - b * r1) * idet, f1 = (a * r1 - bt * r0) * idet;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = g * 4 + q;
            const double old = in_k ? 0.0 : cur[q];
            cur[q] = row == p ? f0 : row == p + 1 ? f1 : fma(-c1[q], f1, fma(-c0[q], f0, old));
        }
    }
}
__device__ __forceinline__ void als5_inv_store(const double (&cur)[4], double* __restrict__ sG) {
    const int lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) sG[(g * 4 + q) * 34 + 16 + c] = cur[q];
}

// stop / mu decisions of an iteration (bit 0: stop, bit 1: mu x 2, bit 2: mu / 2), as in als7: p_res = sqrt(sum_p) / n and d_res =
// mu sqrt(sum_d) / n only feed thresholds, so they are decided on the squared sums (mu is a power of two: mu^2 sum_d is exact) and the IEEE
// sqrt / divide expressions NumPy rounds through are evaluated only when a comparison is closer than 1e-9 to its threshold -- ~100
// dependent instructions less on the path between two phases.  Wave-uniform inputs: every wave takes the same decision.
__device__ __forceinline__ int als5_decide(double sum_p, double sum_d, double mu, int n) {
    const double tol2 = (1e-4 * n) * (1e-4 * n);
    const double sP = sum_p, sD = (mu * mu) * sum_d;
    bool stop = sP < tol2 && sD < tol2, up = sP > 100.0 * sD, down = sD > 100.0 * sP;
    const double eps = 1e-9;
    const bool amb = !(sP > 1e-200) || !(sD > 1e-200) || fabs(sP - tol2) <= eps * tol2 || fabs(sD - tol2) <= eps * tol2 ||
                     fabs(sP - 100.0 * sD) <= eps * sP || fabs(sD - 100.0 * sP) <= eps * sD;
    if (amb) {
        asm volatile("" ::: "memory");   // (keeps this a branch: the IEEE sqrt / divide sequences stay off the common path)
        const double p_res = sqrt(sum_p) / n, d_res = mu * sqrt(sum_d) / n;
        stop = p_res < 1e-4 && d_res < 1e-4; up = p_res > 10 * d_res; down = d_res > 10 * p_res;
    }
    return (stop ? 1 : 0) | (up ? 2 : 0) | (down ? 4 : 0);
}

// One graph (index f of the batch) on a 512-thread workgroup; every thread of the workgroup must call it.
template <typename TW, int NMAX, bool WLDS = true>
__device__ __forceinline__ void als5_graph(Als5Lds<NMAX, WLDS>& L, int f, const TW* __restrict__ W,
                                           const int32_t* __restrict__ gcounts, int G, int ldw,
                                           const double* __restrict__ seed, int seed_len, uint8_t* __restrict__ x_bin,
                                           uint8_t* __restrict__ match_mat, int32_t* __restrict__ labels,
                                           int32_t* __restrict__ n_clusters, int32_t* __restrict__ iters_out,
                                           double* __restrict__ wsym_g = nullptr) {
    static_assert(NMAX == 72, "tile grid: 24 x 18 tiles of 3 x 4 elements");
    constexpr int NT5 = 512, TR = 3, TC = 4, LD = Als5Lds<NMAX, WLDS>::LD, FS = Als5Lds<NMAX, WLDS>::FS, NW5 = NT5 / 64, SOLVER = NW5 - 1;
    double *sX = L.sX, *sA = L.sA, *sB = L.sB, *sH = L.sH, *sG = L.sG, *sGin = L.sGin, *sRed = L.sRed;
    double* sW = WLDS ? L.sW : wsym_g;      // (WLDS = false: global memory, written below and read back by the same threads' workgroup)
    int *sGid = L.sGid, *sKeep = L.sKeep;
    uint8_t* sVis = L.sVis;
    int &s_n = L.s_n, &s_r = L.s_r;
    const int tid = threadIdx.x, wave = tid >> 6;
    __syncthreads();   // the LDS may still be in use by the caller's previous phase
    if (tid == 0) {
        int n = 0, total = 0, gmax = 0;
        for (int g = 0; g < G; ++g) {
            int c = gcounts[f * G + g];
            c = c < 0 ? 0 : c;
            total += c;
            for (int k = 0; k < c && n < NMAX; ++k) sGid[n++] = g;
            if (c > gmax) gmax = c;
        }
        s_n = total;
        const int r = 2 * gmax;
        s_r = r < total ? r : total;
    }
    __syncthreads();
    const int n = s_n, r = s_r;
    int32_t* lab = labels + (size_t)f * ldw;
    if (n == 0 || n > NMAX || n > ldw || r > 16 || n * r > seed_len) {
        for (int i = tid; i < ldw; i += NT5) lab[i] = -1;
        if (tid == 0) { n_clusters[f] = 0; iters_out[f] = (n == 0) ? 0 : -1; }
        return;
    }
    const TW* Wf = W + (size_t)f * ldw * ldw;
    // ---- element tiles ----
    const bool own = tid < 24 * 18;
    // consecutive lanes own consecutive row triples of one column block: a wave reads ~3 distinct B rows (broadcasts) and 24 A rows
    // that the factor stride spreads over the banks
    const int i0 = own ? (tid % 24) * TR : 0, j0 = own ? (tid / 24) * TC : 0;
    double z[TR][TC], y[TR][TC], xp[TR][TC];
    unsigned same_grp = 0;      // bit (a * TC + b): nodes of one group (view / tracklet block), or an element beyond n: Z forced to 0
    unsigned on_diag = 0;       // bit (a * TC + b): i == j: Z forced to 1
    for (int e = tid; e < NMAX * LD; e += NT5) { sX[e] = 0.0; sW[e] = 0.0; }   // rows / columns beyond n are read (times zero factors)
    __syncthreads();
#pragma unroll
    for (int a = 0; a < TR; ++a)
#pragma unroll
        for (int b = 0; b < TC; ++b) {
            const int i = i0 + a, j = j0 + b;
            const bool ok = own && i < n && j < n;
            double wv, x1;
            if constexpr (sizeof(TW) == 4) {
                // float32 affinity: W is symmetrised in float32 and the FIRST X1 = Z - (Y - W + beta) / mu is float32 arithmetic too
                // (NumPy's dtype propagation in mv_association.py:263-277); everything after it is float64
                const float p = ok ? (float)Wf[i * ldw + j] : 0.f, q = ok ? (float)Wf[j * ldw + i] : 0.f;
                const float w32 = fmulr(0.5f, faddr(p, q));
                wv = (double)w32;
                const float qq = faddr(-w32, 0.1f) / 64.f;
                x1 = (double)(w32 - qq);
            } else {
                const double p = ok ? (double)Wf[i * ldw + j] : 0., q = ok ? (double)Wf[j * ldw + i] : 0.;
                wv = 0.5 * (p + q);
                x1 = wv - ((0.0 - wv) + 0.1) / 64.0;
            }
            z[a][b] = wv; xp[a][b] = wv; y[a][b] = 0.0;
            if (ok) { sW[i * LD + j] = wv; sX[i * LD + j] = x1; }
            if (!ok || sGid[i] == sGid[j]) same_grp |= 1u << (a * TC + b);
            if (ok && i == j) on_diag |= 1u << (a * TC + b);
        }
    for (int e = tid; e < NMAX * FS; e += NT5) {
        const int k = e / FS, a = e - k * FS;
        sA[e] = (k < n && a < r) ? seed[k * r + a] : 0.0;
        sB[e] = 0.0;
    }
    // factor-update roles: threads 0..287 = (rank slot fa, block of four columns fj0..fj0+3); the last wave is the solver.
    // (Tried on these phases, all bit-identical: a 2 x 4 output tile per thread on 144 threads -- 40 % fewer LDS bytes, but the two H
    // phases went 14.6 k -> 18.2 k cycles and the whole path 67 k -> 58 k frames/s: 2.25 waves cannot cover the LDS latency; eight rows of
    // X1 per step instead of four -- no change; both together -- 56 k frames/s.)
    // Output tile per worker thread: 2 rank slots x 2 columns (fa, fa + 1; fj0, fj0 + 1) -- the same four outputs per thread as a 1 x 4
    // tile, but one 16-byte read of each operand per k instead of an 8-byte and two 16-byte ones: these phases run at the LDS's
    // bandwidth, and this form moves 20 % fewer bytes in a third fewer instructions.  Every output's sum is unchanged: bit-identical.
    // matrix-core waves of the factor updates: the column blocks 0 .. 4 on the waves 0, 1, 2, 4, 5 -- not on wave 3, which shares its
    // SIMD with the solver wave: the sixteen dependent pivots are the iteration's critical path and run best with the SIMD to themselves
    const bool mw = wave < 3 || wave == 4 || wave == 5;
    const int li = tid & 15, lq = (tid & 63) >> 4, jb = 16 * (wave < 3 ? wave : wave - 1) + li;
    __syncthreads();
    // The solver wave (its 16 dependent pivots are the longest chain of an iteration, so they never stand alone):
    //   phase H   workers: H = A^T X1            | solver: inverts A^T A + rho I (raw Gram matrix from the previous XZY phase + rho)
    //   phase H2  workers: H2 = B^T X1^T         | solver: inverts B^T B + rho I (Gram matrix formed by 256 threads in a short phase)
    //   phase XZY tile owners: X = A B^T, Z, Y   | solver: forms the raw A^T A of the next iteration (rho is added once mu is decided)
    const int n4 = (n + 3) & ~3;      // factor rows beyond n are zero: the products run over whole groups of four rows
    if (wave == SOLVER) als5_gram_mfma<FS>(sA, n4, r, 50.0 / 64.0, true, sGin);
    __syncthreads();
    // the solver wave shares its SIMD with a worker wave and is the younger of the two: at equal priority it only gets the issue slots
    // the worker leaves (MI355X_MICROARCH.md, two waves per SIMD) -- but it is the critical path, the worker waits for it
    if (wave == SOLVER) __builtin_amdgcn_s_setprio(3);

    double mu = 64.0;
    int iters = 1000;
#ifdef MVMC_ALS_PROFILE
    __shared__ long long prof5[16];
    if (tid < 16) prof5[tid] = 0;
    long long _t5 = clock64();
#define A5PROF(k) { const long long _t = clock64(); if ((tid & 63) == 0 && (wave == 0 || wave == SOLVER)) prof5[(wave == SOLVER ? 8 : 0) + (k)] += _t - _t5; _t5 = _t; }
#else
#define A5PROF(k)
#endif
    if (wave == SOLVER) {
        // ---- the solver wave's own loop: the same seven barriers per iteration and the same decisions as the workers' loop below (two
        // loops, so that neither's registers are live in the other's code: the tile state does not spill around the unrolled inversion) ----
        // (Tried and dropped, each within noise of this form or slower: the pivots of inv(A^T A + rho I) spread speculatively over the
        // XZY / X1 / H phases -- they crawl beside the tiles' LDS traffic --; the pivot column through v_readlane instead of the LDS
        // crossbar -- 32 scalar moves per pivot cost more than 16 crossbar reads.)
        // (Tried in round 3: Newton-Schulz on the matrix cores from the previous iteration's inverse, X <- X + X R, R <- R R, eight
        // v_mfma_f64_16x16x4_f64 per step.  Exact to rounding in four or five steps -- but an fp64 MFMA occupies the pipe for ~64 cycles,
        // so a step is ~1 k cycles and the inversion ~6 k, what the sixteen Gauss-Jordan pivots take: 91.8 k -> 92.3 k frames/s.  Dropped.)
        // The inversion for the A update starts as soon as mu is decided -- the solver wave makes the decisions itself right after the
        // residual sums are in -- and so runs beside the tile owners' X1 phase and the workers' H phase instead of beside H alone: its
        // first PRE pivots before the barrier that ends the X1 phase, the rest after it.
        // Round 3: and it starts SPECULATIVELY before that, with the current mu, in the time this wave used to wait for the tile owners at
        // the end of the XZY phase; mu moves in ~40 % of the iterations, and only then the load and the first PRE pivots are redone.
        // Bit-identical either way: the same operations on the same data.  Same-box A/B on config 5 (frames/s): no speculation 106.1 k,
        // PRE = 4 107.6 k, PRE = 8 103.8 k -- the stand-alone phase profile liked 8 better (an iteration 18.5 k -> 17.7 k cycles), the
        // chains do not: the pivots crawl beside the tile waves' LDS traffic and delay the barrier everybody waits at.
        constexpr int PRE = MVMC_ALS5_PRE;
        double cur[4];
        als5_inv_load(cur, sGin, r, 50.0 / mu, false);             // (iteration 0: A^T A + rho I was formed before the loop)
        als5_inv_pivots<0, PRE>(cur);
        for (int it = 0; it < 1000; ++it) {
            const double rho = 50.0 / mu;
            als5_inv_pivots<PRE, 16>(cur);                         // inv(A^T A + rho I), the rest, while the workers form H
            als5_inv_store(cur, sG);
            A5PROF(2)
            __syncthreads();
            A5PROF(3)
            __syncthreads();                                       // (workers: B = inv H)
            A5PROF(4)
            __syncthreads();                                       // (waves 0 .. 2: B^T B in three partial sums over k)
            als5_inv_load(cur, L.sGp[0], r, rho, true, 3);         // inv(B^T B + rho I) while the workers form H2
            als5_inv_pivots<0, 16>(cur);
            als5_inv_store(cur, sG);
            A5PROF(2)
            __syncthreads();
            A5PROF(3)
            __syncthreads();                                       // (workers: A = inv H2)
            A5PROF(4)
            // raw A^T A of the next iteration while the tiles are updated -- at the workers' priority: nothing waits for it before the phase
            // ends, but everybody waits for wave 3, which shares this wave's SIMD and got no issue slots while the product ran at priority 3
            // (the XZY phase lasted 5.7 k cycles where the other tile waves needed 3.5 k)
            __builtin_amdgcn_s_setprio(0);
            als5_gram_mfma<FS>(sA, n4, r, 0.0, false, sGin);
            if ((tid & 63) == 0) { sRed[wave] = 0.0; sRed[8 + wave] = 0.0; }
            MVMC_WAVE_SYNC();
            als5_inv_load(cur, sGin, r, 50.0 / mu, true);          // speculation: mu stays
            als5_inv_pivots<0, PRE>(cur);
            A5PROF(5)
            __syncthreads();
            __builtin_amdgcn_s_setprio(3);
            A5PROF(6)
            double sum_p = 0.0, sum_d = 0.0;
#pragma unroll
            for (int q = 0; q < NW5; ++q) { sum_p += sRed[q]; sum_d += sRed[8 + q]; }
            const int dec = als5_decide(sum_p, sum_d, mu, n);
            if (dec & 1) { iters = it + 1; break; }
            if (dec & 6) {
                mu = (dec & 2) ? 2 * mu : mu / 2;
                als5_inv_load(cur, sGin, r, 50.0 / mu, true);      // A^T A + rho I with the NEW rho: the first pivots beside the X1 phase
                als5_inv_pivots<0, PRE>(cur);
            }
            A5PROF(0)
            __syncthreads();                                       // (tile owners: X1 of the next iteration)
            A5PROF(1)
        }
    } else {
        for (int it = 0; it < 1000; ++it) {
            // (X1 = Z - (Y - W + beta) / mu of this iteration is in sX: written before the loop, then at the end of the previous iteration)
            const double inv_mu = 1.0 / mu;      // mu = 64 * 2^k: the reciprocal is exact, x * inv_mu == x / mu bit for bit
            // ================= B = (inv(A^T A + rho I) (A^T X1))^T =================
            // The products of the factor updates run on the matrix cores (v_mfma_f64_16x16x4_f64, operand layout: als5_gram_mfma): wave
            // cb < 5 makes the sixteen columns 16 cb .. of the 16 x n results -- H = A^T X1 in n / 4 instructions per wave (eighteen at
            // n = 72) with two 8-byte LDS reads each, where the FMA form (288 threads, 2 x 2 output tiles) took 6.6 k cycles per product at
            // the LDS's bandwidth.  Columns beyond n (and beyond the 74 doubles of a row) come out as finite garbage and are not stored.
            if (mw) {
                // (all operands first -- rows beyond n are zero, so the trip count is the constant 18 and the loads are one batch --, then
                // the chain of matrix instructions: with a load pair in front of every instruction the phase took 4.1 k cycles, not 2.6 k)
                als5_d4 acc = {0.0, 0.0, 0.0, 0.0};
                double fa_[NMAX / 4], xb_[NMAX / 4];
#pragma unroll
                for (int u = 0; u < NMAX / 4; ++u) { fa_[u] = sA[(4 * u + lq) * FS + li]; xb_[u] = sX[(4 * u + lq) * LD + jb]; }
#pragma unroll
                for (int u = 0; u < NMAX / 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa_[u], xb_[u], acc, 0, 0, 0);
                if (jb < LD) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) sH[(lq + 4 * v) * LD + jb] = acc[v];
                }
            }
            A5PROF(2)
            __syncthreads();
            A5PROF(3)
            if (mw) {      // B^T = inv(G) H: four instructions per wave
                als5_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int b0 = 0; b0 < 16; b0 += 4)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sG[li * 34 + 16 + b0 + lq], sH[(b0 + lq) * LD + jb], acc, 0, 0, 0);
                if (jb < n) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) sB[jb * FS + lq + 4 * v] = acc[v];
                }
            }
            __syncthreads();
            A5PROF(4)
            // ================= A = (inv(B^T B + rho I) (B^T X1^T))^T =================
            // B^T B: a third of the k range on each of the waves 0 .. 2 (six matrix instructions instead of eighteen on the solver wave, which
            // everybody waited for); the solver wave adds the three partial sums, rho and the identity padding while loading
            if (wave < 3) {
                const int k_lo = wave * (NMAX / 3), k_hi = k_lo + NMAX / 3;
                als5_gram_mfma<FS>(sB, k_hi < n4 ? k_hi : n4, r, 0.0, false, L.sGp[wave], k_lo);
            }
            __syncthreads();
            if (mw) {             // H2[a][i] = sum_k B[k][a] X1[i][k]
                als5_d4 acc = {0.0, 0.0, 0.0, 0.0};
                double fb_[NMAX / 4], xb_[NMAX / 4];
#pragma unroll
                for (int u = 0; u < NMAX / 4; ++u) { fb_[u] = sB[(4 * u + lq) * FS + li]; xb_[u] = sX[jb * LD + 4 * u + lq]; }
#pragma unroll
                for (int u = 0; u < NMAX / 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fb_[u], xb_[u], acc, 0, 0, 0);
                if (jb < LD) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) sH[(lq + 4 * v) * LD + jb] = acc[v];
                }
            }
            A5PROF(2)
            __syncthreads();
            A5PROF(3)
            if (mw) {
                als5_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int b0 = 0; b0 < 16; b0 += 4)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sG[li * 34 + 16 + b0 + lq], sH[(b0 + lq) * LD + jb], acc, 0, 0, 0);
                if (jb < n) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) sA[jb * FS + lq + 4 * v] = acc[v];
                }
            }
            __syncthreads();
            A5PROF(4)

            // ================= X = A B^T ; Z ; Y ; residuals =================
            double acc_p = 0.0, acc_d = 0.0;
            if (own) {
                // (Tried in round 3: X on the matrix cores with the ELEMENT OWNERSHIP in the accumulator layout of v_mfma_f64_16x16x4_f64 --
                // 5 x 5 tiles of 16 x 16 dealt to the seven worker waves, four matrix instructions per tile, Z / Y / previous X / W of a
                // lane's sixteen elements in its registers, the update straight on the accumulators, tiles beyond n skipped: 102 KB of
                // operand reads per iteration instead of 387 KB, exact against the oracle on all 186 graphs.  The stand-alone phase profile
                // liked it -- an iteration 17.7 k -> 17.3 k cycles at n = 72, 18.8 k -> 16.0 k at n = 64 -- the chains did not: 108.2 k ->
                // 106.5 k frames/s on the same box.  An fp64 matrix instruction occupies the SIMD's pipe like the sixteen vector FMAs it
                // stands for (tools/mfma_f64_rate.hip: MFMA waves and vector waves on one SIMD take the SUM of their times), so the tiles'
                // padding -- 25 x 256 elements for 72 x 72 -- is paid in full.  Dropped.)
                // X = A B^T on the 3 x 4 tile in four passes over the rank (four slots each): 14 16-byte loads in flight per pass and 28
                // doubles of operands next to the 48 of state -- more of the rank at once does not fit 256 VGPRs (two waves per SIMD) and
                // a spilled state costs a global-memory round trip per phase.  Elements beyond n have zero factors, zero state and the
                // forced-zero bit: branch-free, they add 0
                double x[TR][TC];
    #pragma unroll
                for (int a = 0; a < TR; ++a)
    #pragma unroll
                    for (int b = 0; b < TC; ++b) x[a][b] = 0.0;
    #pragma unroll
                for (int part = 0; part < 4; ++part) {
                    double2 ar[TR][2], bq[TC][2];
    #pragma unroll
                    for (int a = 0; a < TR; ++a)
    #pragma unroll
                        for (int q = 0; q < 2; ++q) ar[a][q] = *reinterpret_cast<const double2*>(&sA[(i0 + a) * FS + part * 4 + 2 * q]);
    #pragma unroll
                    for (int b = 0; b < TC; ++b)
    #pragma unroll
                        for (int q = 0; q < 2; ++q) bq[b][q] = *reinterpret_cast<const double2*>(&sB[(j0 + b) * FS + part * 4 + 2 * q]);
    #pragma unroll
                    for (int a = 0; a < TR; ++a)
    #pragma unroll
                        for (int b = 0; b < TC; ++b)
                            // (one chain of fused multiply-adds per element: the phase is bound by instruction issue -- two waves per SIMD,
                            // twelve independent chains each --, and the pairwise form it replaces cost six instructions per four terms)
                            x[a][b] = fma(ar[a][1].y, bq[b][1].y, fma(ar[a][1].x, bq[b][1].x, fma(ar[a][0].y, bq[b][0].y, fma(ar[a][0].x, bq[b][0].x, x[a][b]))));
                }
    #pragma unroll
                for (int a = 0; a < TR; ++a)
    #pragma unroll
                    for (int b = 0; b < TC; ++b) {
                        const unsigned bit = 1u << (a * TC + b);
                        const double xv = x[a][b];
                        double zz = xv + y[a][b] * inv_mu;
                        zz = fmin(fmax(zz, 0.0), 1.0);         // (two instructions; the compare / select form of the same clamp took five)
                        zz = (same_grp & bit) ? 0.0 : zz;
                        zz = (on_diag & bit) ? 1.0 : zz;
                        const double dz = xv - zz, dx = xv - xp[a][b];
                        y[a][b] = y[a][b] + mu * dz;
                        z[a][b] = zz;
                        xp[a][b] = xv;
                        acc_p += dz * dz;
                        acc_d += dx * dx;
                    }
            }
            A5PROF(5)
            acc_p = wave_sum_dpp(acc_p);
            acc_d = wave_sum_dpp(acc_d);
            if ((tid & 63) == 0) { sRed[wave] = acc_p; sRed[8 + wave] = acc_d; }
            __syncthreads();
            A5PROF(6)
            double sum_p = 0.0, sum_d = 0.0;
    #pragma unroll
            for (int q = 0; q < NW5; ++q) { sum_p += sRed[q]; sum_d += sRed[8 + q]; }
            const int dec = als5_decide(sum_p, sum_d, mu, n);
            if (dec & 1) { iters = it + 1; break; }
            if (dec & 2) mu = 2 * mu;
            else if (dec & 4) mu = mu / 2;
            // ---- X1 of the next iteration (sX is free: both factor updates are done) ----
            if (own) {
                const double inv_next = 1.0 / mu;
    #pragma unroll
                for (int a = 0; a < TR; ++a)
                    if (i0 + a < n) {
                        const double2 w01 = *reinterpret_cast<const double2*>(&sW[(i0 + a) * LD + j0]);
                        const double2 w23 = *reinterpret_cast<const double2*>(&sW[(i0 + a) * LD + j0 + 2]);
                        const double x0 = z[a][0] - ((y[a][0] - w01.x) + 0.1) * inv_next, x1 = z[a][1] - ((y[a][1] - w01.y) + 0.1) * inv_next;
                        const double x2 = z[a][2] - ((y[a][2] - w23.x) + 0.1) * inv_next, x3 = z[a][3] - ((y[a][3] - w23.y) + 0.1) * inv_next;
                        *reinterpret_cast<double2*>(&sX[(i0 + a) * LD + j0]) = make_double2(x0, x1);
                        *reinterpret_cast<double2*>(&sX[(i0 + a) * LD + j0 + 2]) = make_double2(x2, x3);
                    }
            }
            A5PROF(0)
            __syncthreads();
            A5PROF(1)
        }
    }
    // ---- tail: X_bin = (X + X^T) / 2 > 0.5, closure (k = n-1 only), labels -- same rules as als_kernel ----
    if (wave == SOLVER) __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    if (own) {
#pragma unroll
        for (int a = 0; a < TR; ++a)
#pragma unroll
            for (int b = 0; b < TC; ++b)
                if (i0 + a < n && j0 + b < n) sX[(i0 + a) * LD + j0 + b] = xp[a][b];
    }
    uint8_t* sBin = reinterpret_cast<uint8_t*>(sA);
    uint8_t* sOut = reinterpret_cast<uint8_t*>(sB);
    uint8_t* sTmp = reinterpret_cast<uint8_t*>(sH);
    __syncthreads();
    for (int e = tid; e < n * n; e += NT5) {
        const int i = e / n, j = e - i * n;
        sBin[e] = (0.5 * (sX[i * LD + j] + sX[j * LD + i])) > 0.5;
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += NT5) {
        const int i = e / n, j = e - i * n;
        sOut[e] = 0;
        sTmp[e] = sBin[e] | (sBin[i * n + (n - 1)] & sBin[(n - 1) * n + j]);
    }
    for (int i = tid; i < n; i += NT5) sVis[i] = 0;
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        const bool skip = sVis[i] != 0;
        __syncthreads();
        if (!skip)
            for (int j = tid; j < n; j += NT5)
                if (sTmp[i * n + j]) { sVis[j] = 1; sOut[j * n + i] = 1; }
        __syncthreads();
    }
    for (int c = tid; c < n; c += NT5) {
        int sc = 0;
        for (int j = 0; j < n; ++j) sc += sOut[j * n + c];
        sKeep[c] = sc >= 2;
    }
    __syncthreads();
    for (int row = tid; row < ldw; row += NT5) {
        int label = -1;
        if (row < n) {
            int ord = 0;
            for (int c = 0; c < n; ++c) {
                if (!sKeep[c]) continue;
                if (sOut[row * n + c]) { label = ord; break; }
                ++ord;
            }
        }
        lab[row] = label;
    }
    if (tid == 0) {
        int k = 0;
        for (int c = 0; c < n; ++c) k += sKeep[c];
        n_clusters[f] = k;
        iters_out[f] = iters;
#ifdef MVMC_ALS_PROFILE
        // diagnostic build: cycles per iteration {X1, barrier, H (wave 0) | G + inverse (wave 3), barrier wait, apply + barrier, X/Z/Y, reduce}
        for (int q = 0; q < 16; ++q) lab[q] = (int)(prof5[q] / iters);
#endif
    }
    if (x_bin || match_mat) {
        for (int e = tid; e < ldw * ldw; e += NT5) {
            const int i = e / ldw, j = e - i * ldw;
            const bool in = i < n && j < n;
            if (x_bin) x_bin[(size_t)f * ldw * ldw + e] = in ? sBin[i * n + j] : 0;
            if (match_mat) match_mat[(size_t)f * ldw * ldw + e] = in ? sOut[i * n + j] : 0;
        }
    }
}

template <typename TW, int NMAX>
__global__ void __launch_bounds__(512, 2)
als5_kernel(const TW* __restrict__ W, const int32_t* __restrict__ gcounts, int G, int ldw,
            const double* __restrict__ seed, int seed_len, uint8_t* __restrict__ x_bin,
            uint8_t* __restrict__ match_mat, int32_t* __restrict__ labels, int32_t* __restrict__ n_clusters,
            int32_t* __restrict__ iters_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char als5_lds[];
    auto& L = *reinterpret_cast<Als5Lds<NMAX>*>(als5_lds);
    als5_graph<TW, NMAX>(L, blockIdx.x, W, gcounts, G, ldw, seed, seed_len, x_bin, match_mat, labels, n_clusters, iters_out);
}

// ------------------------------------------------------------------------------------------------
// standalone closure + labelling (transform_closure, mv_association.py:99-121; the cluster rule of
// parse_match_result, motion_capture.py:419-425) for callers that bring their own binary matrix
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
closure_kernel(const uint8_t* __restrict__ x_bin, const int32_t* __restrict__ n_nodes, int ld,
               uint8_t* __restrict__ match_mat, int32_t* __restrict__ labels, int32_t* __restrict__ n_clusters) {
    __shared__ uint8_t sT[MVMC_MAX_NODES * MVMC_MAX_NODES];
    __shared__ uint8_t sO[MVMC_MAX_NODES * MVMC_MAX_NODES];
    __shared__ uint8_t sVis[MVMC_MAX_NODES];
    __shared__ int sKeep[MVMC_MAX_NODES];
    const int f = blockIdx.x, tid = threadIdx.x;
    int n = n_nodes[f];
    n = n < 0 ? 0 : (n > ld ? ld : n);
    const uint8_t* xb = x_bin + (size_t)f * ld * ld;
    for (int e = tid; e < n * n; e += 64) {
        const int i = e / n, j = e - i * n;
        sT[e] = (xb[i * ld + j] != 0) | ((xb[i * ld + (n - 1)] != 0) & (xb[(n - 1) * ld + j] != 0));
        sO[e] = 0;
    }
    for (int i = tid; i < n; i += 64) sVis[i] = 0;
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        const bool skip = sVis[i] != 0;
        __syncthreads();
        if (!skip)
            for (int j = tid; j < n; j += 64)
                if (sT[i * n + j]) { sVis[j] = 1; sO[j * n + i] = 1; }
        __syncthreads();
    }
    for (int c = tid; c < n; c += 64) {
        int s = 0;
        for (int j = 0; j < n; ++j) s += sO[j * n + c];
        sKeep[c] = s >= 2;
    }
    __syncthreads();
    for (int row = tid; row < ld; row += 64) {
        int label = -1;
        if (row < n) {
            int ord = 0;
            for (int c = 0; c < n; ++c) {
                if (!sKeep[c]) continue;
                if (sO[row * n + c]) { label = ord; break; }
                ++ord;
            }
        }
        labels[(size_t)f * ld + row] = label;
    }
    if (tid == 0) {
        int k = 0;
        for (int c = 0; c < n; ++c) k += sKeep[c];
        n_clusters[f] = k;
    }
    if (match_mat)
        for (int e = tid; e < ld * ld; e += 64) {
            const int i = e / ld, j = e - i * ld;
            match_mat[(size_t)f * ld * ld + e] = (i < n && j < n) ? sO[i * n + j] : 0;
        }
}

// ------------------------------------------------------------------------------------------------
// labels -> member lists
// ------------------------------------------------------------------------------------------------
__global__ void members_kernel(const int32_t* __restrict__ labels, const int32_t* __restrict__ counts, int F, int C,
                               int P, int K, int V, int32_t* __restrict__ members, int32_t* __restrict__ n_members) {
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const int N = C * P;
    int32_t* mem = members + (size_t)f * K * V;
    int32_t* nm = n_members + (size_t)f * K;
    for (int k = 0; k < K; ++k) nm[k] = 0;
    for (int e = 0; e < K * V; ++e) mem[e] = -1;
    int node = 0;
    for (int c = 0; c < C; ++c) {
        int cnt = counts[f * C + c];
        cnt = cnt < 0 ? 0 : (cnt > P ? P : cnt);
        for (int p = 0; p < cnt; ++p, ++node) {
            int l = labels[(size_t)f * N + node];
            if (l < 0 || l >= K) continue;
            int m = nm[l]++;
            if (m < V) mem[l * V + m] = (f * C + c) * P + p;
        }
    }
}

}  // namespace

#ifndef MVMC_DEVICE_ONLY   // (mvmc_chain.hip includes the device code above)
// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" int mvmc_ingest(const void* kps, int dtype, int n_frames, int n_views, int p_max, int n_joints_in,
                           const int32_t* counts_in, double min_score, int min_valid, double min_bb_size,
                           double* kps17, int32_t* counts_out, mvmcStream_t stream) {
    if (!kps || !kps17 || !counts_out || n_frames < 0 || n_views <= 0 || p_max <= 0) return MVMC_ERR_ARG;
    if (n_joints_in != 25 && n_joints_in != 17) return MVMC_ERR_ARG;
    if (dtype != MVMC_F32 && dtype != MVMC_F64) return MVMC_ERR_ARG;
    if (n_frames == 0) return MVMC_OK;
    const int nq = n_views * p_max;
    size_t shm = (size_t)nq * 51 * sizeof(double) + (size_t)nq * 2 * sizeof(int);
    if (shm > 64 * 1024) return MVMC_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MVMC_F32)
        hipLaunchKernelGGL(ingest_kernel<float>, dim3(n_frames), dim3(256), shm, s, (const float*)kps, n_views, p_max,
                           n_joints_in, counts_in, min_score, min_valid, min_bb_size, kps17, counts_out);
    else
        hipLaunchKernelGGL(ingest_kernel<double>, dim3(n_frames), dim3(256), shm, s, (const double*)kps, n_views,
                           p_max, n_joints_in, counts_in, min_score, min_valid, min_bb_size, kps17, counts_out);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}

extern "C" int mvmc_fmats(const double* K, const double* Rt, int n_views, float* F, mvmcStream_t stream) {
    if (!K || !Rt || !F || n_views <= 0) return MVMC_ERR_ARG;
    int n = n_views * n_views;
    hipLaunchKernelGGL(fmats_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, K, Rt, n_views, F);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}

extern "C" int mvmc_affinity(const double* kps17, const int32_t* counts, const float* Fmats, int n_frames,
                             int n_views, int p_max, float* D, float* S, mvmcStream_t stream) {
    if (!kps17 || !counts || !Fmats || n_frames < 0 || n_views <= 0 || p_max <= 0) return MVMC_ERR_ARG;
    const int N = n_views * p_max;
    if (N > MVMC_MAX_NODES) return MVMC_ERR_UNSUPPORTED;
    if (n_frames == 0) return MVMC_OK;
    size_t shm = (size_t)N * 51 * sizeof(double) + (size_t)2 * N * N * sizeof(float) + (size_t)2 * N * sizeof(int) + 16;
    hipLaunchKernelGGL(affinity_kernel, dim3(n_frames), dim3(64), shm, (hipStream_t)stream, kps17, counts, Fmats,
                       n_views, p_max, D, S);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}

template <typename TW>
static int launch_als(const TW* W, const int32_t* gc, int F, int G, int n_max, int r_max, const double* seed,
                      int seed_len, uint8_t* xb, uint8_t* mm, int32_t* lab, int32_t* nc, int32_t* it, hipStream_t s) {
#define MVMC_ALS(NM, RM, NT)                                                                              \
    do {                                                                                                  \
        const size_t lds = sizeof(AlsGenLds<NM, RM, NT>);                                                 \
        if (lds > 65536 && hipFuncSetAttribute((const void*)als_kernel<TW, NM, RM, NT>,                   \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return MVMC_ERR_LAUNCH;                                                                       \
        hipLaunchKernelGGL((als_kernel<TW, NM, RM, NT>), dim3(F), dim3(NT), lds, s, W, gc, G, n_max, seed, \
                           seed_len, xb, mm, lab, nc, it);                                                \
    } while (0)
#define MVMC_ALS2(NM)                                                                                     \
    hipLaunchKernelGGL((als2_kernel<TW, NM>), dim3(F), dim3(64), 0, s, W, gc, G, n_max, seed, seed_len, xb, mm, \
                       lab, nc, it)
#define MVMC_ALS4(NM)                                                                                     \
    hipLaunchKernelGGL((als4_kernel<TW, NM>), dim3(F), dim3(256), 0, s, W, gc, G, n_max, seed, seed_len, xb, mm, \
                       lab, nc, it)
    // WHICH VARIANT A CALL REACHES (n = n_max nodes, r = r_max = min(n, 2 g_max), F graphs in the launch) -- every one is reached:
    //   n <= 32, r <= 16, F <= 4096   als4_kernel<., 24 | 32>   one 256-thread workgroup per graph; inside it rank <= 8 and n <= 24 (every
    //     or n <= 24, r <= 8, any F    graph of configs 1-4) run als7_iterate (solver wave + three worker waves), the rest als4_iterate.
    //                                 The temporal path (625 graphs per launch), the chain kernel's SMALL layout (same device functions),
    //                                 config 3 and the all-frames-cold protocol (10 k graphs of n = 20, rank 8)
    //   n <= 32, r <= 16, F >  4096   als2_kernel<., 24 | 32>   one wave per graph: many graphs with more than four people per view or
    //     (and n > 24 or r > 8)        more than 24 nodes (als4_iterate's 14 k cycles per iteration on 512 resident graphs lose against
    //                                 1,024 resident waves there)
    //   n <= 32, r >  16              als_kernel<., 24 | 32, r = n>   (a view with more than 8 people)
    //   n <= 72, r <= 16, F <= 4096   als5_kernel<., 72>        512 threads per graph, products on the matrix cores: config 5's temporal
    //                                 graphs (64 poses + 8 tracklets); the chain kernel's BIG layout runs the same als5_graph
    //   n <= 64 | 80, r <= 16         als_kernel<., 64 | 80, 16>   config 5 in the many-graphs form (match_spatial of 25 k frames)
    //   n <= 80, r <= 32              als_kernel<., 80, 32>     the repair tier (tracker.T_WIDE = 16 tracklet slots)
    // Variants are sized by (max nodes, max rank).  r_max is only the caller's bound (2 x largest group
    // capacity); the kernels check the frame's actual rank and flag iters = -1 if it does not fit.
    // Few graphs per launch (the temporal path: one frame of every chain): the launch lasts as long as its
    // slowest graph, so a whole workgroup works on each; many graphs: one wave per graph fills the machine.
    // (round 4: graphs the solver-wave form als7_iterate holds -- n <= 24, rank <= 8 -- take the workgroup kernel at ANY batch size: 512
    // graphs resident at 7.4 k cycles per iteration beat als2's 1,024 at ~20 k; config 3, 10 k graphs: 484 k -> 642 k frames/s)
    const bool few = F <= 4096 || (n_max <= 24 && r_max <= 8);
    if (n_max <= 24 && r_max <= 16) { if (few) MVMC_ALS4(24); else MVMC_ALS2(24); }
    else if (n_max <= 32 && r_max <= 16) { if (few) MVMC_ALS4(32); else MVMC_ALS2(32); }
    else if (n_max <= 24) MVMC_ALS(24, 24, 64);
    else if (n_max <= 32) MVMC_ALS(32, 32, 128);
    else if (n_max <= 72 && r_max <= 16 && few) {   // config 5 (C8 P8, + 8 tracklets): the workgroup form the chain kernel's BIG layout runs
        const size_t lds = sizeof(Als5Lds<72>);
        if (hipFuncSetAttribute((const void*)als5_kernel<TW, 72>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return MVMC_ERR_LAUNCH;
        hipLaunchKernelGGL((als5_kernel<TW, 72>), dim3(F), dim3(512), lds, s, W, gc, G, n_max, seed, seed_len, xb, mm, lab, nc, it);
    }
    else if (n_max <= 64 && r_max <= 16) MVMC_ALS(64, 16, 256);
    else if (n_max <= 80 && r_max <= 16) MVMC_ALS(80, 16, 512);
    else if (n_max <= 80) MVMC_ALS(80, 32, 512);   // up to 16 groups members (the repair tier's 16 tracklet slots): 121 KB of LDS
    else return MVMC_ERR_UNSUPPORTED;
#undef MVMC_ALS
#undef MVMC_ALS2
#undef MVMC_ALS4
    return MVMC_OK;
}

extern "C" int mvmc_als_associate(const void* W, int w_dtype, const int32_t* group_counts, int n_frames,
                                  int n_groups, int n_max, int g_max, const double* seed_table, int seed_len, uint8_t* x_bin,
                                  uint8_t* match_mat, int32_t* labels, int32_t* n_clusters, int32_t* iters,
                                  mvmcStream_t stream) {
    if (!W || !group_counts || !seed_table || !labels || !n_clusters || !iters) return MVMC_ERR_ARG;
    if (n_frames < 0 || n_groups <= 0 || n_max <= 0 || n_max > MVMC_MAX_NODES) return MVMC_ERR_ARG;
    if (w_dtype != MVMC_F32 && w_dtype != MVMC_F64) return MVMC_ERR_ARG;
    if (n_frames == 0) return MVMC_OK;
    if (g_max <= 0) return MVMC_ERR_ARG;
    int r_max = 2 * g_max < n_max ? 2 * g_max : n_max;  // rank = min(n, 2 * largest group)
    hipStream_t s = (hipStream_t)stream;
    int st = (w_dtype == MVMC_F32)
                 ? launch_als<float>((const float*)W, group_counts, n_frames, n_groups, n_max, r_max, seed_table,
                                     seed_len, x_bin, match_mat, labels, n_clusters, iters, s)
                 : launch_als<double>((const double*)W, group_counts, n_frames, n_groups, n_max, r_max, seed_table,
                                      seed_len, x_bin, match_mat, labels, n_clusters, iters, s);
    if (st != MVMC_OK) return st;
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}

extern "C" int mvmc_cluster_members(const int32_t* labels, const int32_t* counts, int n_frames, int n_views,
                                    int p_max, int k_max, int v_max, int32_t* members, int32_t* n_members,
                                    mvmcStream_t stream) {
    if (!labels || !counts || !members || !n_members || n_views <= 0 || p_max <= 0 || k_max <= 0 || v_max <= 0)
        return MVMC_ERR_ARG;
    if (n_frames <= 0) return n_frames == 0 ? MVMC_OK : MVMC_ERR_ARG;
    hipLaunchKernelGGL(members_kernel, dim3((n_frames + 63) / 64), dim3(64), 0, (hipStream_t)stream, labels, counts,
                       n_frames, n_views, p_max, k_max, v_max, members, n_members);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}

extern "C" int mvmc_closure_labels(const uint8_t* x_bin, const int32_t* n_nodes, int n_frames, int n_max,
                                   uint8_t* match_mat, int32_t* labels, int32_t* n_clusters, mvmcStream_t stream) {
    if (!x_bin || !n_nodes || !labels || !n_clusters || n_max <= 0 || n_max > MVMC_MAX_NODES) return MVMC_ERR_ARG;
    if (n_frames <= 0) return n_frames == 0 ? MVMC_OK : MVMC_ERR_ARG;
    hipLaunchKernelGGL(closure_kernel, dim3(n_frames), dim3(64), 0, (hipStream_t)stream, x_bin, n_nodes, n_max,
                       match_mat, labels, n_clusters);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}
#endif  // MVMC_DEVICE_ONLY
