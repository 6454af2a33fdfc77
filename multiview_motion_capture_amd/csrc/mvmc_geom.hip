// Geometry kernels for gfx950: batched multi-view DLT (TR-1/TR-2) and forward kinematics (FK-1/FK-2).
#include <cstdlib>
#include "mvmc_common.h"
#include "mvmc_postopt.h"

// ------------------------------------------------------------------------------------------------
// DLT: one thread per (problem, joint).  The 2V x 4 system is reduced to its 4x4 normal matrix in
// fp64 registers and the null vector is the eigenvector of the smallest eigenvalue (== last right
// singular vector of A, mv_math_util.py:235-236): inverse iteration on L D L^T; where the spectral gap is
// small or a pivot vanishes, the smallest eigenvalue by cyclic Jacobi and the iteration shifted to it.
// ------------------------------------------------------------------------------------------------
// Upper triangle of a symmetric 4 x 4 matrix: entry (r, c), r <= c, at index U4(r, c) of ten doubles.
__host__ __device__ constexpr int U4(int r, int c) { return r <= c ? r * 4 - r * (r - 1) / 2 + (c - r) : c * 4 - c * (c - 1) / 2 + (r - c); }
// One Jacobi rotation (P, Q) of the cyclic sweep, EIGENVALUES ONLY (no eigenvector accumulation: ten doubles of state where the matrix
// pair of the first version held thirty-two -- that fallback alone took the kernel from 82 to 164 VGPRs, i.e. from six waves per SIMD
// to three; the eigenvector now comes from two or three inverse iterations shifted to the eigenvalue found here).
template <int P, int Q>
__device__ __forceinline__ void jacobi_rot4_ev(double (&b)[10]) {
    const double apq = b[U4(P, Q)];
    if (fabs(apq) < 1e-300) return;
    const double theta = (b[U4(Q, Q)] - b[U4(P, P)]) / (2.0 * apq);
    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k == P || k == Q) continue;
        const double akp = b[U4(k, P)], akq = b[U4(k, Q)];
        b[U4(k, P)] = c * akp - s * akq;
        b[U4(k, Q)] = s * akp + c * akq;
    }
    b[U4(P, P)] -= t * apq;
    b[U4(Q, Q)] += t * apq;
    b[U4(P, Q)] = 0.0;
}

// One triangulated point from the views `get(v, kp, Pc)` hands out (v = 0 .. V-1; false = no such member): kp <- {x, y, score},
// Pc -> the view's 3x4 projection.  out[0..2] = X, out[3] = mean score of the views used; NaN when the cluster is empty.
// mv_math_util.py:152-187 (triangulate_point_groups_from_multiple_views_linear) + :215-240 (the DLT of one point).
// VU > 0: the view loops are unrolled VU times behind `v < V` tests (V <= VU), for callers whose `get` serves view v out of registers.
template <int VU = 0, typename Get>
__device__ __forceinline__ void dlt_point(int V, double min_score, Get get, double* __restrict__ o) {
    // upper triangle of the normal matrix A^T A (rows r1 = x P_3 - P_1, r2 = y P_3 - P_2 of every view used).  ONE pass over the views in
    // the common case: the views with score >= min_score are accumulated while all views are counted; only a point that fewer than
    // two such views see ("< 2 valid views -> resort to all views", mv_math_util.py:177-182) is accumulated again over all of them.
    // (The first version counted in a pass of its own: the members, slots and keypoints of every view were read twice.)
    double a00, a01, a02, a03, a11, a12, a13, a22, a23, a33;
    double ssum = 0.0;
    int nused = 0, n_all = 0, n_ok = 0;
    auto accumulate = [&](bool use_all) {
    a00 = 0.0; a01 = 0.0; a02 = 0.0; a03 = 0.0; a11 = 0.0; a12 = 0.0; a13 = 0.0; a22 = 0.0; a23 = 0.0; a33 = 0.0;
    ssum = 0.0; nused = 0; n_all = 0; n_ok = 0;
    auto view = [&](int v) {
        double kp[3]; const double* Pc;
        if (!get(v, kp, Pc)) return;
        const double x = kp[0], y = kp[1], sc = kp[2];
        ++n_all;
        const bool ok = sc >= min_score;
        n_ok += ok ? 1 : 0;
        if (!use_all && !ok) return;
        const double p0 = x * Pc[8] - Pc[0], p1 = x * Pc[9] - Pc[1], p2 = x * Pc[10] - Pc[2], p3 = x * Pc[11] - Pc[3];
        const double q0 = y * Pc[8] - Pc[4], q1 = y * Pc[9] - Pc[5], q2 = y * Pc[10] - Pc[6], q3 = y * Pc[11] - Pc[7];
        // two fused multiply-adds per entry (the sum p p + q q + a in one chain: a third fewer instructions than product, fma, add)
        a00 = fma(p0, p0, fma(q0, q0, a00)); a01 = fma(p0, p1, fma(q0, q1, a01)); a02 = fma(p0, p2, fma(q0, q2, a02));
        a03 = fma(p0, p3, fma(q0, q3, a03)); a11 = fma(p1, p1, fma(q1, q1, a11)); a12 = fma(p1, p2, fma(q1, q2, a12));
        a13 = fma(p1, p3, fma(q1, q3, a13)); a22 = fma(p2, p2, fma(q2, q2, a22)); a23 = fma(p2, p3, fma(q2, q3, a23));
        a33 = fma(p3, p3, fma(q3, q3, a33));
        ssum += sc;
        ++nused;
    };
    if constexpr (VU > 0) {
#pragma unroll
        for (int v = 0; v < VU; ++v)
            if (v < V) view(v);
    } else {
        for (int v = 0; v < V; ++v) view(v);
    }
    };
    accumulate(false);
    if (n_all == 0) {
        const double nan = __longlong_as_double(0x7ff8000000000000LL);
        o[0] = o[1] = o[2] = o[3] = nan;
        return;
    }
    const bool use_all = n_ok < 2;
    if (use_all) accumulate(true);
    // The right singular vector of the smallest singular value (mv_math_util.py:152-160: SVD of the 2 nv x 4 system, last row of V^T) =
    // the eigenvector of the smallest eigenvalue of the normal matrix a.  Inverse iteration on a = L D L^T started from e4: the first
    // iterate is L^-T e4, i.e. the inhomogeneous least-squares point (X, 1); every further solve multiplies the error by
    // lambda_min / lambda_2 (~1e-5 for pixel noise against a real baseline), so three to five solves reach 1e-13 -- ~300 flops where
    // the cyclic Jacobi sweeps this replaces took ~3,000 and left the kernel ALU bound at 13 TFLOP/s.  A point
    // seen by fewer than two views has a rank-deficient matrix (a pivot vanishes): the Jacobi path below keeps handling those.
    const double tr = a00 + a11 + a22 + a33;
    double e0 = 0.0, e1 = 0.0, e2 = 0.0, e3 = 1.0;
    // inverse iteration on (a - sigma I) = L D L^T from the start vector e; false = a pivot vanished or the iterates did not settle
    // (reciprocals by v_rcp_f64 + two Newton steps, ~1 ulp: the factors and the normalisation only steer an iteration whose fixed point
    // does not depend on them, and an IEEE division is ~28 dependent instructions -- a dozen of them were 40 % of a point's instructions)
    // -> 0 settled, 1 not settled after max_it solves, 2 a pivot vanished (no solve was made: e is untouched)
    auto inverse_iteration = [&](double sigma, int max_it) -> int {
        const double floor_ = 1e-13 * tr;
        const double d0 = a00 - sigma, i0 = fast_rcp64(d0);
        const double l10 = a01 * i0, l20 = a02 * i0, l30 = a03 * i0;
        const double d1 = (a11 - sigma) - l10 * a01, i1 = fast_rcp64(d1);
        const double l21 = (a12 - l20 * a01) * i1, l31 = (a13 - l30 * a01) * i1;
        const double d2 = (a22 - sigma) - l20 * a02 - l21 * l21 * d1, i2 = fast_rcp64(d2);
        const double l32 = (a23 - l30 * a02 - l31 * l21 * d1) * i2;
        double d3 = (a33 - sigma) - l30 * a03 - l31 * l31 * d1 - l32 * l32 * d2;
        if (!(d0 > floor_ && d1 > floor_ && d2 > floor_)) return 2;
        // (the last pivot is ~lambda_min - sigma: rounding may push it to zero or below for consistent observations; its size only scales
        // the iterates, their direction comes from L)
        const double tiny = 1e-30 * tr + 1e-300;
        if (!(d3 > tiny)) d3 = tiny;
        const double i3 = fast_rcp64(d3);
        double x0 = e0, x1 = e1, x2 = e2, x3 = e3;
        bool conv = false;
        double ch_prev = 1.0;
        for (int it = 0; it < max_it; ++it) {
            // L y = x;  z = y / D;  L^T w = z
            const double y0 = x0, y1 = x1 - l10 * y0, y2 = x2 - l20 * y0 - l21 * y1, y3 = x3 - l30 * y0 - l31 * y1 - l32 * y2;
            const double w3 = y3 * i3;
            const double w2 = y2 * i2 - l32 * w3;
            const double w1 = y1 * i1 - l21 * w2 - l31 * w3;
            const double w0 = y0 * i0 - l10 * w1 - l20 * w2 - l30 * w3;
            // normalised by the component of largest magnitude (sign included): converged iterates repeat
            double m = w0;
            if (fabs(w1) > fabs(m)) m = w1;
            if (fabs(w2) > fabs(m)) m = w2;
            if (fabs(w3) > fabs(m)) m = w3;
            const double inv = fast_rcp64(m);   // (the bare v_rcp_f64 will not do although the factor only scales the iterate: its error,
                                                // ~1e-8 and not a smooth function of m, keeps consecutive iterates 1e-9 apart for ever:
                                                // every point then fell through to the Jacobi path, 6.3 ms instead of 1.2)
            const double n0 = w0 * inv, n1 = w1 * inv, n2 = w2 * inv, n3 = w3 * inv;
            const double ch = fmax(fmax(fabs(n0 - x0), fabs(n1 - x1)), fmax(fabs(n2 - x2), fabs(n3 - x3)));
            x0 = n0; x1 = n1; x2 = n2; x3 = n3;
            // settled: the iterate repeats to 1e-13 -- or, the changes shrinking geometrically (by lambda_min / lambda_2 per solve), the
            // NEXT change would: ch (ch / ch_prev) <= 1e-13 with the ratio itself below 1e-3.  The second test saves the solve that
            // only confirms (three solves instead of four at the usual gap of ~1e-5); the iterate it stops at is within that product of
            // the fixed point.
            if (it > 0 && (ch <= 1e-13 || (it > 1 && ch <= 1e-3 * ch_prev && ch * ch <= 1e-13 * ch_prev))) { conv = true; break; }
            ch_prev = ch;
        }
        e0 = x0; e1 = x1; e2 = x2; e3 = x3;
        return conv ? 0 : 1;
    };
    // not settled after eight solves = a small spectral gap (clusters of mismatched poses, gross outliers: lambda_min / lambda_2 > ~0.03;
    // 15 % of the points of the Shelf clusters, none of the synthetic ones), or a vanished pivot: the smallest eigenvalue by cyclic
    // Jacobi on the upper triangle, then the same iteration shifted to just below it -- the error then shrinks by
    // 1e-14 tr / (lambda_2 - lambda_min) per solve, whatever the ratio of the two
    if (inverse_iteration(0.0, 8) != 0) {
        double b[10] = {a00, a01, a02, a03, a11, a12, a13, a22, a23, a33};
        for (int sweep = 0; sweep < 16; ++sweep) {
            const double off = b[1] * b[1] + b[2] * b[2] + b[3] * b[3] + b[5] * b[5] + b[6] * b[6] + b[8] * b[8];
            if (off <= 1e-36 * tr * tr) break;
            jacobi_rot4_ev<0, 1>(b); jacobi_rot4_ev<0, 2>(b); jacobi_rot4_ev<0, 3>(b);
            jacobi_rot4_ev<1, 2>(b); jacobi_rot4_ev<1, 3>(b); jacobi_rot4_ev<2, 3>(b);
        }
        const double lmin = fmin(fmin(b[0], b[4]), fmin(b[7], b[9]));
        accumulate(use_all);   // (the matrix again, from the views: keeping it live across the sweeps would cost twenty registers of the hot path's budget)
        e0 = 0.0; e1 = 0.0; e2 = 0.0; e3 = 1.0;
        // A pivot that vanishes even in the shifted matrix = the null space has more than one dimension (a point one view sees, two
        // views on one line of sight): the reference returns whichever null vector LAPACK happens to produce, there is no point to
        // agree on -- NaN, never the start vector (0, 0, 0, 1) dressed up as the point (0, 0, 0)
        const int st = inverse_iteration(lmin - 1e-14 * tr, 8);
        if (st == 2 || !(e3 == e3)) { e0 = e1 = e2 = 0.0; e3 = 0.0; }   // (0 / 0 below gives NaN)
    }
    const double rw = 1.0 / e3;     // (one IEEE division; the three quotients differ from e / e3 by an ulp at most)
    o[0] = e0 * rw; o[1] = e1 * rw; o[2] = e2 * rw;
    o[3] = ssum / (double)nused;
}

__global__ void __launch_bounds__(256)
dlt_kernel(const double* __restrict__ kps17, const double* __restrict__ Pm, const int32_t* __restrict__ members,
           int B, int V, int C, int Pmax, int J, double min_score, double* __restrict__ out) {
    // the projection matrices through LDS (C <= 16): every thread reads twelve doubles per view, the same for all lanes of a cluster
    __shared__ double sP[16 * 12];
    const bool lds_p = C <= 16;
    if (lds_p) for (int e = threadIdx.x; e < C * 12; e += blockDim.x) sP[e] = Pm[e];
    __syncthreads();
    const double* Pv = lds_p ? sP : Pm;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * J) return;
    const int b = idx / J, j = idx - b * J;
    const int ps = J * 3;  // pose stride
    const int32_t* mem = members + (size_t)b * V;
    dlt_point(V, min_score, [&](int v, double (&kp)[3], const double*& Pc) {
        const int q = mem[v];
        if (q < 0) return false;
        const double* k3 = kps17 + (size_t)q * ps + j * 3;
        kp[0] = k3[0]; kp[1] = k3[1]; kp[2] = k3[2];
        Pc = Pv + ((q / Pmax) % C) * 12;
        return true;
    }, out + (size_t)idx * 4);
}

// ------------------------------------------------------------------------------------------------
// Ingest + DLT in one pass (BASELINE config 2: triangulation only).  A 256-thread workgroup takes G frames at a time (G chosen by the
// launcher so that the G x K x 17 points of the group fill its threads: 15 frames at C5 P1), persistent over the frame groups:
//   A  the group's raw keypoints (f32 or f64, OpenPose-25 or COCO-17), one (pose, joint) triple per thread and trip, into LDS as the
//      17-joint f64 poses mvmc_ingest would write;
//   B  filter_bad_pose per pose, C  per-view compaction (pose_def.py:262-270, motion_capture.py:1023-1043: the rule of mvmc_ingest);
//   D  one thread per (frame, cluster, joint): the DLT of that point from LDS.
// The 17-joint tensor (2 KB per view-frame, written and read again by the two-kernel form) never exists in HBM: per frame
// 12 C P J bytes in, 17 x 32 bytes per cluster out.  (A first version gave every frame its own wave: 17 of 64 lanes at work in D, and
// fp64 vector instructions cost the same whether 17 or 64 lanes are on -- 9.7 ms for 2 M frames against 6.8 ms for the two kernels.)
// members (F, K, V): pose indices in mvmc_ingest's output numbering, (f C + c) P + slot, all of frame f; -1 = none.
// ------------------------------------------------------------------------------------------------
#ifndef MVMC_DLT_WAVES
#define MVMC_DLT_WAVES 4
#endif
template <typename T>
__global__ void __launch_bounds__(256, MVMC_DLT_WAVES)
ingest_dlt_kernel(const T* __restrict__ kps, int F, int G, int C, int P, int J_in, const int32_t* __restrict__ counts_in,
                  double min_score_in, int min_valid, double min_bb, const double* __restrict__ Pm,
                  const int32_t* __restrict__ members, int K, int V, double min_score, double* __restrict__ out,
                  int32_t* __restrict__ counts_out) {
    // The poses stay in LDS in the INPUT type (float -> double is exact, so widening at the read gives what mvmc_ingest's f64 tensor
    // holds): half the LDS bytes for f32 input, twice the workgroups per CU.
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
    const int tid = threadIdx.x, nq = C * P;
    T* pose = reinterpret_cast<T*>(sm_raw);                                                  // [G][nq][17][3]
    int* keep = reinterpret_cast<int*>(sm_raw + (((size_t)G * nq * 51 * sizeof(T) + 7) & ~(size_t)7));   // [G][nq]
    int* src_of = keep + G * nq;                                    // [G][nq] ingest slot -> source pose of the raw layout, -1 = empty
    int* cnt_l = src_of + G * nq;                                   // [G][C]    the group's view counts
    int* mem_l = cnt_l + G * C;                                     // [G][K][V] the group's cluster members
    __shared__ double sP[16 * 12];                                  // the projection matrices (C <= 16 checked by the launcher)
    for (int e = tid; e < C * 12; e += 256) sP[e] = Pm[e];
    const int n_groups = (F + G - 1) / G;
    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int f0 = grp * G, g_n = min(G, F - f0);
        const T* src = kps + (size_t)f0 * nq * J_in * 3;
        // Stage A: EVERYTHING the group reads from global memory, issued as one batch of independent loads -- keypoints four trips at a
        // time, the view counts, the cluster members -- then one wait.  (The first version looked the joint map up in constant memory in
        // front of every keypoint load, waited for each trip's loads before the next trip's, and read the members and counts from global
        // memory inside the later stages: ~20 dependent memory round trips per group, which is what the kernel's 2.7 ms were made of.)
        const int n_tr = g_n * nq * 17;
        constexpr int NB = 4;                                        // trips per batch (five tie, six spill the staging array)
        for (int t0 = tid; t0 < n_tr; t0 += 256 * NB) {              // t = (frame in group, pose, COCO joint)
            T v3[NB][3];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int t = t0 + 256 * u;
                if (t < n_tr) {
                    const int gq = t / 17, j = t - gq * 17;
                    const int js = (J_in == 25) ? op25_to_coco17(j) : j;
                    const T* s3 = src + ((size_t)gq * J_in + js) * 3;
                    v3[u][0] = s3[0]; v3[u][1] = s3[1]; v3[u][2] = s3[2];
                }
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int t = t0 + 256 * u;
                if (t < n_tr) {
                    T* d3 = pose + (size_t)t * 3;
                    d3[0] = v3[u][0]; d3[1] = v3[u][1]; d3[2] = v3[u][2];
                }
            }
        }
        for (int t = tid; t < g_n * C; t += 256) cnt_l[t] = counts_in ? counts_in[(size_t)f0 * C + t] : P;
        for (int t = tid; t < g_n * K * V; t += 256) mem_l[t] = members[(size_t)f0 * K * V + t];
        __syncthreads();
        for (int t = tid; t < g_n * nq; t += 256) {
            const int g = t / nq, q = t - g * nq, c = q / P, p = q - c * P;
            const int cnt = cnt_l[g * C + c];
            int ok = 0;
            if (p < cnt) {
                int nv = 0;
                double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
                const T* ps = pose + (size_t)t * 51;
                for (int j = 0; j < 17; ++j) {
                    const double kx = (double)ps[j * 3], ky = (double)ps[j * 3 + 1], ks = (double)ps[j * 3 + 2];
                    if (ks > min_score_in) {
                        ++nv;
                        x0 = fmin(x0, kx); x1 = fmax(x1, kx);
                        y0 = fmin(y0, ky); y1 = fmax(y1, ky);
                    }
                }
                ok = (nv >= min_valid) && !((x1 - x0) < min_bb || (y1 - y0) < min_bb);
            }
            keep[t] = ok;
        }
        __syncthreads();
        for (int t = tid; t < g_n * C; t += 256) {
            const int g = t / C, c = t - g * C;
            const int* kp = keep + g * nq + c * P;
            int* so = src_of + g * nq + c * P;
            int k = 0;
            for (int p = 0; p < P; ++p)
                if (kp[p]) so[k++] = c * P + p;
            if (counts_out) counts_out[(f0 + g) * C + c] = k;
            for (; k < P; ++k) so[k] = -1;
        }
        __syncthreads();
        for (int t = tid; t < g_n * K * 17; t += 256) {             // t = (frame in group, cluster, joint)
            const int gk = t / 17, j = t - gk * 17, g = gk / K;
            const int* mem = mem_l + gk * V;
            const int base = (f0 + g) * nq;
            const int* so = src_of + g * nq;
            const T* pg = pose + (size_t)g * nq * 51 + j * 3;
            dlt_point(V, min_score, [&](int v, double (&kp)[3], const double*& Pc) {
                const int d = mem[v] - base;
                if (d < 0 || d >= nq) return false;      // (-1, or a member of another frame: not this kernel's contract)
                const int q = so[d];
                if (q < 0) return false;
                const T* k3 = pg + q * 51;
                kp[0] = (double)k3[0]; kp[1] = (double)k3[1]; kp[2] = (double)k3[2];
                Pc = sP + (d / P) * 12;
                return true;
            }, out + ((size_t)f0 * K * 17 + t) * 4);
        }
        __syncthreads();   // the next group overwrites the LDS block
    }
}

// ------------------------------------------------------------------------------------------------
// The same pass for float32 input as a producer / consumer pipeline inside the workgroup (round 4; what BASELINE config 2 runs).
// Two LDS buffers.  Of a workgroup's four waves ONE is the loader and three triangulate:
//   loader   issues the NEXT group's keypoints as LDS-DMA (global_load_lds_dwordx3: a lane hands the hardware the address of one
//            (x, y, score) triple and the wave's 64 triples land side by side in LDS, 16 bytes apart -- the OpenPose-25 -> COCO-17
//            gather is the lanes' source addresses; no staging registers, no ds_write), waits for them, runs filter_bad_pose and the
//            per-view compaction on them (a lane per (frame, view)) and writes the frame's view counts;
//   the others  one thread per (frame, cluster, joint) of the CURRENT group: the DLT of that point out of LDS, two 16-byte stores.
// ONE barrier per group (raw s_barrier behind s_waitcnt lgkmcnt(0): a __syncthreads() fences vmcnt(0) and would make the triangulating
// waves wait for their result stores): the triangulating waves never execute a load, a filter or a wait for memory.  Which wave loads
// rotates with the workgroup's index, so that the (light) loaders of the workgroups sharing a CU sit on different SIMDs.
// The DMA is inline asm on purpose: hipcc counts a __builtin_amdgcn_global_load_lds and, unable to tell which LDS bytes it writes,
// puts s_waitcnt vmcnt(0) in front of the next LDS read of ANY object (seen in the ISA of the first version).
// Results are those of ingest_dlt_kernel<float> and of mvmc_ingest + mvmc_dlt, bit for bit (same dlt_point on the same numbers).
// History of the round (2 M frames, C5 P1): first version (four stages behind barriers, register staging) 1.52 ms; LDS-DMA double
// buffering with every wave doing everything 1.25 ms; this form: see DESIGN.md section 6.
// ------------------------------------------------------------------------------------------------
#ifndef ID3_GENERIC_VIEWS
#define ID3_GENERIC_VIEWS 1  // the generic view loop for every shape.  0 = up to five views unrolled with member -> slot -> keypoint of all
                             // views read in three rounds before the arithmetic (round 4's form: 126 registers against 107).  Measured in
                             // round 5 on the same box, 2 M frames at C5 P1, float32 points: 1.102 ms unrolled, 1.073 ms generic -- with the
                             // loads out of the triangulating waves since round 4, the fifteen staged operands only cost registers
#endif
#ifndef ID3_WPC
#define ID3_WPC 4            // workgroups per CU the kernel is built for (registers: launch bounds; LDS: the buffer sizes below)
#endif
#ifndef ID3_TRIPLES_N
#define ID3_TRIPLES_N 960
#define ID3_MISC_N 256
#define ID3_SLOT_N 128
#define ID3_SP_VIEWS 16
#endif
constexpr int ID3_DLT_THREADS = 192;     // three triangulating waves
constexpr int ID3_TRIPLES = ID3_TRIPLES_N;         // per buffer: G frames x nq poses x 17 joints (11 frames at C5 P1), FOUR floats each: the
                                         // hardware writes a lane's 12 bytes at 16-byte lane stride (tools/glds12_test.hip)
constexpr int ID3_MISC_INTS = ID3_MISC_N;       // view counts [G][C], then cluster members [G][K][V]
constexpr int ID3_SLOT_INTS = ID3_SLOT_N;       // ingest slot -> source pose [G][nq]
struct Id3Buf {
    __attribute__((aligned(16))) float pose[ID3_TRIPLES * 4];
    int small[ID3_MISC_INTS];
    int src_of[ID3_SLOT_INTS];
};

// a pointer the compiler must hold in scalar registers (the "s" operand of the statements below: a value it cannot prove uniform it
// would otherwise hand over in vector registers, which is not an encoding of these instructions)
template <typename T>
__device__ __forceinline__ const T* id3_uniform(const T* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const T*>(((unsigned long long)hi << 32) | lo);
}
// One LDS-DMA of 12 bytes per lane: lane l's triple at base + off lands at LDS byte address lds_wave + 16 l (a 4-byte hole behind each).
// M0 is written in the statement that reads it and restored (the compiler reserves it).
__device__ __forceinline__ void id3_glds12(const float* base, unsigned off, unsigned lds_wave) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(base), "s"(__builtin_amdgcn_readfirstlane(lds_wave)) : "memory");
}
// one word per lane (lane stride 4 bytes): the small per-group tables
__device__ __forceinline__ void id3_glds4(const int32_t* base, unsigned off, unsigned lds_wave) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(base), "s"(__builtin_amdgcn_readfirstlane(lds_wave)) : "memory");
}
__device__ __forceinline__ unsigned id3_lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}

struct Id3Args {
    const float* kps; const double* Pm; int F, G, C, P, J_in; const int32_t* counts_in; double min_score_in; int min_valid; double min_bb;
    const int32_t* members; int K, V; double min_score; void* out; int32_t* counts_out;   // out: (F,K,17,4) f64, or f32 in the OUT32 instance
};

// the loader wave: group grp -> buffer b (DMA, wait, filter + compaction)
__device__ __forceinline__ void id3_load_group(const Id3Args& A, Id3Buf& b, int grp, float score_thr) {
    const int lane = threadIdx.x & 63;
    const int nq = A.C * A.P, C = A.C, P = A.P, f0 = grp * A.G, g_n = min(A.G, A.F - f0);
    const float* src = id3_uniform(A.kps + (size_t)f0 * nq * A.J_in * 3);
    const int n_tr = g_n * nq * 17;                                     // elements = (frame in group, pose, COCO joint) triples
    const unsigned pose0 = id3_lds_addr(b.pose);
#ifdef ID3_NO_LOAD      // (timing experiment: the keypoints and slots of a workgroup's first two groups are re-used)
    const bool stale = grp >= 2 * (int)gridDim.x;
#else
    const bool stale = false;
#endif
    if (!stale)
    for (int t0 = 0; t0 < n_tr; t0 += 64) {
        const int t = t0 + lane;
        if (t < n_tr) {
            const int gq = t / 17, j = t - gq * 17;
            const int js = (A.J_in == 25) ? op25_to_coco17(j) : j;
            id3_glds12(src, (unsigned)((gq * A.J_in + js) * 12), pose0 + t0 * 16);
        }
    }
    const int n_c = g_n * C, n_m = g_n * A.K * A.V;
    const unsigned small0 = id3_lds_addr(b.small);
    for (int t0 = 0; t0 < n_c; t0 += 64) {
        const int t = t0 + lane;
        if (t < n_c) {
            if (A.counts_in) id3_glds4(id3_uniform(A.counts_in + (size_t)f0 * C), t * 4u, small0 + t0 * 4);
            else b.small[t] = P;
        }
    }
    for (int t0 = 0; t0 < n_m; t0 += 64) {
        const int t = t0 + lane;
        if (t < n_m) id3_glds4(id3_uniform(A.members + (size_t)f0 * A.K * A.V), t * 4u, small0 + (A.G * C + t0) * 4);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");        // this wave's DMA has landed (and its earlier count stores are out)
    if (stale) return;
    // filter_bad_pose + per-view compaction (motion_capture.py:1023-1043; pose_def.py:262-270: the rule of mvmc_ingest), in float32,
    // exactly: score_thr is the smallest float above the double threshold, minima / maxima of floats are floats, and the box sides are
    // differenced in double
    for (int t = lane; t < n_c; t += 64) {
        const int g = t / C, c = t - g * C;
        const int cnt = b.small[t];
        int* so = b.src_of + g * nq + c * P;
        int k = 0;
        for (int p = 0; p < P; ++p) {
            if (p >= cnt) break;
            int nv = 0;
            float x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
            const float4* ps = reinterpret_cast<const float4*>(b.pose) + (size_t)(g * nq + c * P + p) * 17;
#pragma unroll
            for (int j = 0; j < 17; ++j) {
                const float4 k4 = ps[j];
                const bool on = k4.z >= score_thr;
                nv += on ? 1 : 0;
                x0 = fminf(x0, on ? k4.x : INFINITY); x1 = fmaxf(x1, on ? k4.x : -INFINITY);
                y0 = fminf(y0, on ? k4.y : INFINITY); y1 = fmaxf(y1, on ? k4.y : -INFINITY);
            }
            const double wx = (double)x1 - (double)x0, wy = (double)y1 - (double)y0;
            if ((nv >= A.min_valid) && !(wx < A.min_bb || wy < A.min_bb)) so[k++] = c * P + p;
        }
        if (A.counts_out) A.counts_out[(f0 + g) * C + c] = k;
        for (; k < P; ++k) so[k] = -1;
    }
}

// a triangulating thread: point dt = (frame in group, cluster, joint) of group grp out of buffer b
template <int VU, bool OUT32>
__device__ __forceinline__ void id3_point(const Id3Args& A, const Id3Buf& b, const double* __restrict__ sP, int grp, int dt,
                                          unsigned p_magic) {
    const int nq = A.C * A.P, K = A.K, V = A.V, f0 = grp * A.G, g_n = min(A.G, A.F - f0);
    if (dt >= g_n * K * 17) return;
#ifdef ID3_NO_DLT       // (timing experiment)
    return;
#endif
    const int gk = dt / 17, j = dt - gk * 17, g = gk / K;
    const int* mem = b.small + A.G * A.C + gk * V;
    const int base = (f0 + g) * nq;
    const int* so = b.src_of + g * nq;
    const float4* pg = reinterpret_cast<const float4*>(b.pose) + (size_t)g * nq * 17 + j;
    // OUT32: the point leaves as ONE 16-byte store (x, y, z, score as float32: SURVEY 8(d)'s 16 P J bytes per frame) -- the values
    // are the float64 result rounded once, at the store
    double r4[4];
    double* o = OUT32 ? r4 : reinterpret_cast<double*>(A.out) + ((size_t)f0 * K * 17 + dt) * 4;
    if constexpr (VU > 0) {
        // member -> slot -> keypoint of ALL views first: three rounds of independent LDS reads instead of three dependent reads per view
        int dd[VU], qq[VU];
        float kx[VU], ky[VU], ks[VU];
#pragma unroll
        for (int v = 0; v < VU; ++v) dd[v] = v < V ? mem[v] - base : -1;
#pragma unroll
        for (int v = 0; v < VU; ++v) qq[v] = (dd[v] >= 0 && dd[v] < nq) ? so[dd[v]] : -1;      // (-1, or a member of another frame: not this kernel's contract)
#pragma unroll
        for (int v = 0; v < VU; ++v) {
            const float4 k4 = pg[(qq[v] < 0 ? 0 : qq[v]) * 17];
            kx[v] = k4.x; ky[v] = k4.y; ks[v] = k4.z;
        }
        dlt_point<VU>(V, A.min_score, [&](int v, double (&kp)[3], const double*& Pc) {
            if (qq[v] < 0) return false;
            kp[0] = (double)kx[v]; kp[1] = (double)ky[v]; kp[2] = (double)ks[v];
            Pc = sP + ((unsigned)dd[v] * p_magic >> 16) * 12;        // d / P (d < C P <= 128: exact)
            return true;
        }, o);
    } else {
        dlt_point(V, A.min_score, [&](int v, double (&kp)[3], const double*& Pc) {
            const int d = mem[v] - base;
            if (d < 0 || d >= nq) return false;
            const int q = so[d];
            if (q < 0) return false;
            const float4 k4 = pg[q * 17];
            kp[0] = (double)k4.x; kp[1] = (double)k4.y; kp[2] = (double)k4.z;
            Pc = sP + ((unsigned)d * p_magic >> 16) * 12;
            return true;
        }, o);
    }
    if constexpr (OUT32)
        reinterpret_cast<float4*>(A.out)[(size_t)f0 * K * 17 + dt] = make_float4((float)r4[0], (float)r4[1], (float)r4[2], (float)r4[3]);
}

template <int VU, bool OUT32>
__global__ void __launch_bounds__(256, ID3_WPC)
ingest_dlt3_kernel(Id3Args A) {
    __shared__ Id3Buf bufA;
    __shared__ Id3Buf bufB;
    __shared__ double sP[ID3_SP_VIEWS * 12];                                  // the projection matrices (C <= 16 checked by the launcher)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < A.C * 12; e += 256) sP[e] = A.Pm[e];
    const int n_groups = (A.F + A.G - 1) / A.G, stride = gridDim.x;
#ifndef ID3_ROT_SHIFT
#define ID3_ROT_SHIFT 8
#endif
    // rotates with the workgroup's index so that the loaders of the workgroups sharing a CU sit on different SIMDs: with 256 CUs the
    // workgroups b, b + 256, b + 512, b + 768 of the resident grid share one (measured against rotating by b itself: DESIGN.md)
    const int load_wave = (int)((blockIdx.x >> ID3_ROT_SHIFT) & 3u);
    const bool loader = wave == load_wave;
    const int dt = (((wave - load_wave - 1) & 3) << 6) | (tid & 63);   // thread index among the 192 triangulating threads
    const unsigned p_magic = (65536u + (unsigned)A.P - 1u) / (unsigned)A.P;   // d / P = d * p_magic >> 16 for d < 128, P <= 16
    float score_thr = (float)A.min_score_in;                       // k.z > min_score_in (double)  <=>  k.z >= score_thr (float)
    if ((double)score_thr <= A.min_score_in) score_thr = nextafterf(score_thr, INFINITY);
    int grp = blockIdx.x;
    __builtin_amdgcn_s_waitcnt(0x0F70);                            // (the matrices' loads: nothing of the compiler's is pending from here on)
    if (loader && grp < n_groups) id3_load_group(A, bufA, grp, score_thr);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (; grp < n_groups; grp += 2 * stride) {
        // group grp out of bufA while grp + stride is prepared in bufB, then the other way round
        if (loader) { if (grp + stride < n_groups) id3_load_group(A, bufB, grp + stride, score_thr); }
        else id3_point<VU, OUT32>(A, bufA, sP, grp, dt, p_magic);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (grp + stride >= n_groups) break;
        if (loader) { if (grp + 2 * stride < n_groups) id3_load_group(A, bufA, grp + 2 * stride, score_thr); }
        else id3_point<VU, OUT32>(A, bufB, sP, grp + stride, dt, p_magic);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

// ------------------------------------------------------------------------------------------------
// post-optimise of triangulated points (mv_math_util.py:189-210): one wave per problem
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
postopt_kernel(const double* __restrict__ kps, const double* __restrict__ Pm, const int32_t* __restrict__ members,
               int V, int C, int Pmax, int J, double* __restrict__ pts) {
    extern __shared__ double sm[];
    const int b = blockIdx.x, lane = threadIdx.x;
    double* sObs = sm;                 // [nv][J][3]
    double* sP = sm + V * J * 3;       // [nv][12]
    __shared__ int s_nv;
    __shared__ int s_q[64];
    if (lane == 0) {
        int nv = 0;
        for (int v = 0; v < V && nv < 64; ++v) {
            const int q = members[(size_t)b * V + v];
            if (q >= 0) s_q[nv++] = q;
        }
        s_nv = nv;
    }
    __syncthreads();
    const int nv = s_nv;
    if (nv < 1) return;
    for (int e = lane; e < nv * J * 3; e += 64) {
        const int v = e / (J * 3), r = e - v * J * 3;
        sObs[e] = kps[(size_t)s_q[v] * J * 3 + r];
    }
    for (int e = lane; e < nv * 12; e += 64) {
        const int v = e / 12, r = e - v * 12;
        sP[e] = Pm[(size_t)((s_q[v] / Pmax) % C) * 12 + r];
    }
    __syncthreads();
    double X[3] = {0, 0, 0};
    double* o = pts + ((size_t)b * J + (lane < J ? lane : 0)) * 4;
    if (lane < J) { X[0] = o[0]; X[1] = o[1]; X[2] = o[2]; }
    postopt::post_optimize_wave(X, sObs + (lane < J ? lane : 0) * 3, J * 3, sP, nv, J);
    if (lane < J) { o[0] = X[0]; o[1] = X[1]; o[2] = X[2]; }
}

// ------------------------------------------------------------------------------------------------
// FK: R_j = Rx Ry Rz through the reference's quaternion path (axis scaled by 1/(1+1e-10),
// Quaternions.py:444), chained 4x4 products in index order (parents precede children).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
fk_kernel(SkelDev sk, const double* __restrict__ params, int B, double* __restrict__ joints, double* __restrict__ Gout) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int np = 57 + sk.n_side;
    const double* x = params + (size_t)b * np;
    double Gm[18][12];  // rows 0..2 of each global 4x4 (row 3 is 0 0 0 1)
    for (int j = 0; j < 18; ++j) {
        double R[9], t[3];
        euler_to_rot(x + 3 + 3 * j, R);
        if (j == 0) {
            t[0] = x[0]; t[1] = x[1]; t[2] = x[2];
        } else {
            const double len = x[57 + sk.side_map[j]];
            t[0] = sk.dirs[j][0] * len; t[1] = sk.dirs[j][1] * len; t[2] = sk.dirs[j][2] * len;
        }
        if (j == 0) {
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) Gm[0][r * 4 + c] = R[r * 3 + c];
                Gm[0][r * 4 + 3] = t[r];
            }
        } else {
            const double* Gp = Gm[sk.parents[j]];
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c)
                    Gm[j][r * 4 + c] = Gp[r * 4] * R[c] + Gp[r * 4 + 1] * R[3 + c] + Gp[r * 4 + 2] * R[6 + c];
                Gm[j][r * 4 + 3] = Gp[r * 4] * t[0] + Gp[r * 4 + 1] * t[1] + Gp[r * 4 + 2] * t[2] + Gp[r * 4 + 3];
            }
        }
    }
    for (int j = 0; j < 18; ++j) {
        for (int r = 0; r < 3; ++r) joints[((size_t)b * 18 + j) * 3 + r] = Gm[j][r * 4 + 3];
        if (Gout) {
            double* g = Gout + ((size_t)b * 18 + j) * 16;
            for (int e = 0; e < 12; ++e) g[e] = Gm[j][e];
            g[12] = 0; g[13] = 0; g[14] = 0; g[15] = 1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" int mvmc_dlt(const double* kps17, const double* Pmats, const int32_t* members, int n_problems,
                        int v_max, int n_views, int p_max, int n_joints, double min_score, double* out,
                        mvmcStream_t stream) {
    if (!kps17 || !Pmats || !members || !out || v_max <= 0 || n_views <= 0 || p_max <= 0 || n_joints <= 0)
        return MVMC_ERR_ARG;
    if (n_problems <= 0) return n_problems == 0 ? MVMC_OK : MVMC_ERR_ARG;
    const int total = n_problems * n_joints;
    hipLaunchKernelGGL(dlt_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, kps17, Pmats, members,
                       n_problems, v_max, n_views, p_max, n_joints, min_score, out);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}

static int ingest_dlt_launch(const void* kps, int dtype, int n_frames, int n_views, int p_max, int n_joints_in,
                             const int32_t* counts_in, double ingest_min_score, int min_valid, double min_bb_size,
                             const double* Pmats, const int32_t* members, int k_max, int v_max, double min_score, void* out_any, bool out32,
                             int32_t* counts_out, mvmcStream_t stream) {
    double* out = reinterpret_cast<double*>(out_any);
    if (!kps || !Pmats || !members || !out || n_frames < 0 || n_views <= 0 || p_max <= 0 || k_max <= 0 || v_max <= 0) return MVMC_ERR_ARG;
    if (n_joints_in != 25 && n_joints_in != 17) return MVMC_ERR_ARG;
    if (dtype != MVMC_F32 && dtype != MVMC_F64) return MVMC_ERR_ARG;
    if (n_views > 16) return MVMC_ERR_UNSUPPORTED;
    if (n_frames == 0) return MVMC_OK;
    const int nq = n_views * p_max;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MVMC_F32 && !getenv("MVMC_INGEST_DLT_V1") && k_max * 17 <= ID3_DLT_THREADS) {
        // the pipelined kernel: frames per group = what the three triangulating waves take in one trip (11 frames at C5 P1: 187 points)
        // and one LDS buffer holds
        int G3 = ID3_DLT_THREADS / (k_max * 17);
        while (G3 > 0 && ((long long)G3 * nq * 17 > ID3_TRIPLES || (long long)G3 * (n_views + (long long)k_max * v_max) > ID3_MISC_INTS ||
                          G3 * nq > ID3_SLOT_INTS)) --G3;
        if (G3 > n_frames) G3 = n_frames;
        if (G3 > 0 && nq <= 128 && p_max <= 16 && n_views <= ID3_SP_VIEWS) {
            Id3Args A{(const float*)kps, Pmats, n_frames, G3, n_views, p_max, n_joints_in, counts_in, ingest_min_score, min_valid, min_bb_size,
                      members, k_max, v_max, min_score, out, counts_out};
            long long blocks = ((long long)n_frames + G3 - 1) / G3;
            const long long cap = (long long)cus * ID3_WPC;      // persistent and resident (35 KB of LDS each): the workgroups stride over the groups
            if (blocks > cap) blocks = cap;
            if (out32) {
                if (v_max <= 5 && !ID3_GENERIC_VIEWS) hipLaunchKernelGGL((ingest_dlt3_kernel<5, true>), dim3((unsigned)blocks), dim3(256), 0, s, A);
                else hipLaunchKernelGGL((ingest_dlt3_kernel<0, true>), dim3((unsigned)blocks), dim3(256), 0, s, A);
            } else {
                if (v_max <= 5 && !ID3_GENERIC_VIEWS) hipLaunchKernelGGL((ingest_dlt3_kernel<5, false>), dim3((unsigned)blocks), dim3(256), 0, s, A);
                else hipLaunchKernelGGL((ingest_dlt3_kernel<0, false>), dim3((unsigned)blocks), dim3(256), 0, s, A);
            }
            MVMC_CHECK_LAUNCH();
            return MVMC_OK;
        }
    }
    if (out32) return MVMC_ERR_UNSUPPORTED;     // float32 results: the pipelined kernel's shapes only (float32 input, k_max * 17 <= 192, ...)
    const size_t per_frame = (size_t)nq * 51 * (dtype == MVMC_F32 ? 4 : 8) + ((size_t)2 * nq + n_views + (size_t)k_max * v_max) * sizeof(int);
    // frames per group: TWO full trips of the 256 threads through the DLT stage (30 frames at C5 P1: the stages' barriers and the
    // 75-thread filter / compaction stages are per group, and with 15 frames they were a sixth of the time: 1.71 -> 1.49 ms per 2 M
    // frames; 45 frames need 48 KB of LDS and cost the fourth workgroup per CU: 1.68 ms), inside 36 KB of LDS (four workgroups per CU
    // at 124 VGPRs)
    const int g0 = 256 / (k_max * 17) > 0 ? 256 / (k_max * 17) : 1;
    int G = 2 * g0;
    if ((size_t)G * per_frame > 36 * 1024) G = g0;
    while (G > 1 && (size_t)G * per_frame > 36 * 1024) --G;
    if ((size_t)G * per_frame > 64 * 1024) return MVMC_ERR_UNSUPPORTED;
    if (G > n_frames) G = n_frames;
    const size_t shm = (((size_t)G * per_frame + 15) & ~(size_t)15) + 16;
    long long blocks = ((long long)n_frames + G - 1) / G;
    const long long cap = (long long)cus * 16;         // persistent: the workgroups stride over the frame groups
    if (blocks > cap) blocks = cap;
    if (dtype == MVMC_F32)
        hipLaunchKernelGGL(ingest_dlt_kernel<float>, dim3((unsigned)blocks), dim3(256), shm, s, (const float*)kps, n_frames, G, n_views,
                           p_max, n_joints_in, counts_in, ingest_min_score, min_valid, min_bb_size, Pmats, members, k_max, v_max,
                           min_score, out, counts_out);
    else
        hipLaunchKernelGGL(ingest_dlt_kernel<double>, dim3((unsigned)blocks), dim3(256), shm, s, (const double*)kps, n_frames, G,
                           n_views, p_max, n_joints_in, counts_in, ingest_min_score, min_valid, min_bb_size, Pmats, members, k_max,
                           v_max, min_score, out, counts_out);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}

extern "C" int mvmc_ingest_dlt(const void* kps, int dtype, int n_frames, int n_views, int p_max, int n_joints_in,
                               const int32_t* counts_in, double ingest_min_score, int min_valid, double min_bb_size,
                               const double* Pmats, const int32_t* members, int k_max, int v_max, double min_score, double* out,
                               int32_t* counts_out, mvmcStream_t stream) {
    return ingest_dlt_launch(kps, dtype, n_frames, n_views, p_max, n_joints_in, counts_in, ingest_min_score, min_valid, min_bb_size, Pmats,
                             members, k_max, v_max, min_score, out, false, counts_out, stream);
}

extern "C" int mvmc_ingest_dlt_f32(const float* kps, int n_frames, int n_views, int p_max, int n_joints_in,
                                   const int32_t* counts_in, double ingest_min_score, int min_valid, double min_bb_size,
                                   const double* Pmats, const int32_t* members, int k_max, int v_max, double min_score, float* out,
                                   int32_t* counts_out, mvmcStream_t stream) {
    return ingest_dlt_launch(kps, MVMC_F32, n_frames, n_views, p_max, n_joints_in, counts_in, ingest_min_score, min_valid, min_bb_size,
                             Pmats, members, k_max, v_max, min_score, out, true, counts_out, stream);
}

extern "C" int mvmc_fk(const mvmcSkeleton* skel_host, const double* params, int n_problems, double* joints,
                       double* G, mvmcStream_t stream) {
    if (!skel_host || !params || !joints) return MVMC_ERR_ARG;
    if (n_problems <= 0) return n_problems == 0 ? MVMC_OK : MVMC_ERR_ARG;
    SkelDev sk;
    if (!skel_to_dev(skel_host, &sk)) return MVMC_ERR_ARG;
    hipLaunchKernelGGL(fk_kernel, dim3((n_problems + 63) / 64), dim3(64), 0, (hipStream_t)stream, sk, params,
                       n_problems, joints, G);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}

extern "C" int mvmc_triangulate_postopt(const double* kps, const double* Pmats, const int32_t* members, int n_problems,
                                        int v_max, int n_views, int p_max, int n_joints, double* pts,
                                        mvmcStream_t stream) {
    if (!kps || !Pmats || !members || !pts || v_max <= 0 || v_max > 64 || n_views <= 0 || p_max <= 0) return MVMC_ERR_ARG;
    if (n_joints <= 0 || n_joints > 64) return MVMC_ERR_UNSUPPORTED;
    if (n_problems <= 0) return n_problems == 0 ? MVMC_OK : MVMC_ERR_ARG;
    const size_t shm = ((size_t)v_max * n_joints * 3 + (size_t)v_max * 12) * sizeof(double);
    if (shm > 60 * 1024) return MVMC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(postopt_kernel, dim3(n_problems), dim3(64), shm, (hipStream_t)stream, kps, Pmats, members, v_max,
                       n_views, p_max, n_joints, pts);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}
