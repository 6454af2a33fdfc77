// The Krylov tridiagonalisation of mvmc_eigh_tri.h on ONE wave (mvmc_ik1.hip: one wave per solve, no workgroup
// barriers).  Lane l owns column l of the symmetric matrix and keeps all N rows in registers: a[i] = M[i][l].
// Per step: the reflector of row k (one wave reduction), then p = tau M v and the rank-2 update with the row-side
// values v_i, w_i read back from LDS as broadcasts (two values per ds_read_b128) -- three FMAs per live row and
// lane and no cross-lane VALU traffic in the inner loops.  Rows <= k are dead (never read again; their diagonal
// entry survives because v vanishes there) and are skipped in chunks.  The Householder vectors STAY IN THE MATRIX
// REGISTERS: row k is dead after step k and is left untouched from then on (w is forced to zero on the lanes <= k, which
// only ever fed dead entries), so v_k = {1 at lane k+1, a[k] * sc_k beyond} can be re-formed from a[k] and the scalar
// sc_k, which lane k keeps (scv; tau_k likewise in tauv).  apply_q_packed applies Q from those registers (packed two rows to one) -- the common
// path makes its trust-region trial inside the same function, before the registers die, and no reflector ever travels
// to memory; dump_reflectors writes them to the global scratch (hh[k * 64 + lane] for lane > k) for the eigenbasis
// fallback only, whose trials outlive the registers (apply_q_w1).
#pragma once
#include "mvmc_common.h"
#include "mvmc_eigh_tri.h"

namespace eightri {

#define MVMC_ROWS50(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) \
    X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32) X(33) X(34) X(35) X(36) X(37)  \
    X(38) X(39) X(40) X(41) X(42) X(43) X(44) X(45) X(46) X(47) X(48) X(49)

// a[k] for a wave-uniform k (register arrays need static indices)
template <int N>
__device__ __forceinline__ double row_of(const double (&a)[N], int k) {
    double x = 0.0;
#define MVMC_ROWCASE(Z) case Z: if constexpr (Z < N) x = a[Z]; break;
    switch (k) { MVMC_ROWS50(MVMC_ROWCASE) default: break; }
#undef MVMC_ROWCASE
    // The value is pinned here: where its only use sits behind a condition (if (lane == k) d[k] = x), the optimiser sinks the cases'
    // loads into that block as ONE load through a phi of POINTERS to the rows, before the array has been promoted to registers -- the
    // whole matrix then lives in scratch memory (seen in the 30-row instance: 118 scratch stores, 262 loads in the model function)
    asm volatile("" : "+v"(x));
    return x;
}

//   a[N]: the matrix (destroyed); gj: g[lane]; n <= N active size; d, e, tau, v0: LDS, 64 doubles each;
//   vb, pb: LDS exchange vectors, 64 doubles each, 16-byte aligned; out4 = {beta0, tau0, |M|_1, coupling}.
// Same stopping rules as tridiag_krylov: returns the size kk of the leading (range) block (d[0..kk], e[0..kk-1],
// tau[0..kk-1) valid; d[kk] is the first diagonal entry of the null block), n if no sub-diagonal collapses, and -1 if
// one collapses in front of a block that is not null -- the tridiagonalisation is then complete (all n rows) and
// meant for tri_eigh_w1.
template <int N>
__device__ __forceinline__ int tridiag_krylov_w1(double (&a)[N], double gj, int n, double* d, double* e, double* tau, double* v0,
                                                 double* vb, double* pb, double* out4, double& scv, double& tauv, int& ksteps,
                                                 long long* tprof = nullptr) {
    static_assert(N % 2 == 0 && N <= 50, "row count");
    const int lane = threadIdx.x & 63;
    // (Round 6, bit-identical and measured on the chain kernel: the two sums of a step over the rows of 16 lanes that can hold anything
    // -- wave_sum_rows<(N + 15) / 16> -- take four instructions off each reduction, and the 40-row model then carries 79 scratch loads
    // instead of 60: 550 k -> 540 k frames/s.  The full-wave sum stays here; apply_q_packed below takes the short one.)
    auto wave_max = [](double v) { return wave_max_dpp(v); };
    auto lane_value = [](double v, int src) {  // v of lane src (uniform) as a scalar
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
    };
    // (branch-free: the scalars are selected at the end, so that the chain stays in one basic block with whatever the caller overlaps it with)
    auto reflector = [&](double alpha, double sig, double& tk, double& beta, double& sc) {
        const double q2 = alpha * alpha + sig;
        double rs = __builtin_amdgcn_rsq(q2);
        rs = rs * (1.5 - 0.5 * q2 * rs * rs);
        rs = rs * (1.5 - 0.5 * q2 * rs * rs);
        const double nrm = q2 * rs;
        const double b1 = alpha >= 0.0 ? -nrm : nrm;
        const double t1 = 1.0 - alpha * fast_rcp64(b1);
        const double s1 = fast_rcp64(alpha - b1);
        const bool live = sig > 0.0;
        tk = live ? t1 : 0.0; beta = live ? b1 : alpha; sc = live ? s1 : 0.0;
    };
    // M <- H M H with H = I - tk v v^T; rows > k only, in chunks of CH rows behind one wave-uniform test each (a
    // test per row would put a full LDS round trip in front of every FMA; v vanishes on the dead rows of a live chunk)
    // Also tried: the product started on the UNSCALED row before the reflector's scalars are known (M v = M e_{k+1} + sc M x~, all rows in
    // one basic block so that the scheduler can run it inside the latency of the reduction + rsqrt / reciprocal chain): the scheduler
    // hoists every broadcast load to the top and the matrix rows spill (145 scratch stores / 496 loads in the 50-row model, IK 37 -> 122 M
    // cycles per chain); it would need a hand-placed schedule (sched_group_barrier per sub-step of the chain).  Not done.
    // And: the loop software-pipelined across the step boundary -- row, reduction and reflector scalars of step k + 1 (they need row k + 1
    // only) made between the first live chunk of step k's rank-2 update and the rest of it, the next chunk's broadcasts issued in front
    // of that chain.  Bit-identical, but the preloaded operands (40 registers in the 40-row model) and the second set of step scalars do
    // not fit: 63 - 73 scratch stores in the model functions, IK 34.5 -> 84 M cycles per chain; without the preload the 50-row model
    // still spills (36 stores) and nothing is hidden.  Dropped.
    // Batch sizes: what a step waits for is the LDS round trip of the row-side broadcasts (> 100 cycles each; the FMAs of a row are 4 - 8),
    // so the 40-row model takes as many rows per round trip as its registers hold.  Tried and dropped: a software pipeline over 5-row
    // chunks with two or three chunks of operands in flight (loads of chunk c + 2 in front of the FMAs of chunk c) -- more round trips
    // and branches than it hides, IK 37.9 -> 40.1 M cycles per chain, and the 50-row model spilled v_j / w_j.
#ifdef MVMC_TRI_HB
    constexpr int HB = MVMC_TRI_HB;
#else
    constexpr int HB = N <= 40 ? 5 : 3;   // pairs of rows per batch of the rank-2 update
#endif
    // (CHB: rows per round trip of the product.  Twenty for the 40-row model measured best while a wave's latency was the limit; with the
    // chip's VALU and LDS pipes as the limit -- two launches in flight, every slot taken -- the ten dead rows a 20-row chunk drags along on
    // average cost more than the round trip they save: 454.0 k -> 456.1 k frames/s, bit-identical)
#ifdef MVMC_TRI_CHB
    constexpr int CH = 10, CHB = MVMC_TRI_CHB;
#else
    constexpr int CH = 10, CHB = 10;
#endif
    static_assert(N % CH == 0 && N % CHB == 0, "rows per chunk");
#ifdef MVMC_TRI_PROFILE   // diagnostic: cycles of a step's four sections into tprof[0, 3, 4, 6] (tools/tri_step_profile.py)
    long long _tt = clock64();
#define TRSTAMP(k) { const long long _t = clock64(); if (lane == 0 && tprof) tprof[k] += _t - _tt; _tt = _t; }
#else
#define TRSTAMP(k)
#endif
    auto rank2 = [&](int k, double vj, double wj) {
#pragma unroll
        for (int c = 0; c < N; c += CH)
            if (c + CH - 1 > k) {
                // two batches of loads per chunk (6 + 4 rows): ten 16-byte broadcasts in flight at once would push the
                // N = 50 instance over its register budget
#pragma unroll
                for (int h0 = 0; h0 < CH / 2; h0 += HB) {
                    double2 v2[HB], w2[HB];
#pragma unroll
                    for (int u = 0; u < HB; ++u)
                        if (h0 + u < CH / 2) {
                            v2[u] = *reinterpret_cast<const double2*>(&vb[c + 2 * (h0 + u)]);
                            w2[u] = *reinterpret_cast<const double2*>(&pb[c + 2 * (h0 + u)]);
                        }
#pragma unroll
                    for (int u = 0; u < HB; ++u)
                        if (h0 + u < CH / 2) {
                            const int i = c + 2 * (h0 + u);
                            a[i] = fma(-w2[u].x, vj, fma(-v2[u].x, wj, a[i]));
                            a[i + 1] = fma(-w2[u].y, vj, fma(-v2[u].y, wj, a[i + 1]));
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        MVMC_WAVE_SYNC();  // vb / pb are rewritten by the next step
        TRSTAMP(6)   // rank-2 update
    };
    auto two_sided = [&](int k, double tk, double vj) {
        TRSTAMP(0)   // row, reductions and the reflector's scalars
        vb[lane] = vj;
        MVMC_WAVE_SYNC();
        double p0 = 0.0, p1 = 0.0;
        // (the product takes CHB rows per LDS round trip)
#pragma unroll
        for (int c = 0; c < N; c += CHB)
            if (c + CHB - 1 > k) {
#ifdef MVMC_TRI_PB   // (pairs of rows per batch of loads inside a chunk: the 128-register build has no room for a chunk's ten operands at once)
                constexpr int PB = MVMC_TRI_PB;
#pragma unroll
                for (int u0 = 0; u0 < CHB / 2; u0 += PB) {
                    double2 v2[PB];
#pragma unroll
                    for (int u = 0; u < PB; ++u)
                        if (u0 + u < CHB / 2) v2[u] = *reinterpret_cast<const double2*>(&vb[c + 2 * (u0 + u)]);
#pragma unroll
                    for (int u = 0; u < PB; ++u)
                        if (u0 + u < CHB / 2) { p0 += a[c + 2 * (u0 + u)] * v2[u].x; p1 += a[c + 2 * (u0 + u) + 1] * v2[u].y; }
                    __builtin_amdgcn_sched_barrier(0);
                }
#else
                double2 v2[CHB / 2];
#pragma unroll
                for (int u = 0; u < CHB / 2; ++u) v2[u] = *reinterpret_cast<const double2*>(&vb[c + 2 * u]);
#pragma unroll
                for (int u = 0; u < CHB / 2; ++u) { p0 += a[c + 2 * u] * v2[u].x; p1 += a[c + 2 * u + 1] * v2[u].y; }
#endif
            }
        TRSTAMP(3)   // v broadcast + matrix-vector product
        const double p = tk * (p0 + p1);
        const double h = 0.5 * tk * wave_sum_dpp(p * vj);
        const double wj = lane > k ? p - h * vj : 0.0;   // (lanes <= k: dead columns; zero keeps the dead rows intact)
        pb[lane] = wj;
        MVMC_WAVE_SYNC();
        TRSTAMP(4)   // h reduction, w, w broadcast
        rank2(k, vj, wj);
    };
    double anorm;
    {
        double cs = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) cs += fabs(a[i]);
        anorm = uni(wave_max(cs));
    }
    const double tol_c = uni(1e-8 * anorm), tol_n = uni(1e-13 * anorm);   // (long-lived wave-uniform scalars: scalar registers)
    double coupling = 0.0;
    {   // first reflector: H_g g = beta0 e_1
        const double x = lane < n ? gj : 0.0;
        const double alpha = lane_value(x, 0);
        const double sig = wave_sum_dpp(lane > 0 ? x * x : 0.0);
        double tk, beta, sc;
        reflector(alpha, sig, tk, beta, sc);
        const double v = lane == 0 ? 1.0 : x * sc;
        v0[lane] = v;
        if (lane == 0) { out4[0] = beta; out4[1] = tk; out4[2] = anorm; }
        if (tk != 0.0) two_sided(-1, tk, v);
    }
    int kk = n;
    bool unclean = false;
    scv = 0.0; tauv = 0.0; ksteps = 0;
    for (int k = 0; k < n - 1; ++k) {
        const int j1 = k + 1;
        const double x = row_of<N>(a, k);
        if (lane == k) d[k] = x;
        const double alpha = lane_value(x, j1);
        const double sig = wave_sum_dpp((lane > j1 && lane < n) ? x * x : 0.0);
        double tk, beta, sc;
        reflector(alpha, sig, tk, beta, sc);
        const double v = lane == j1 ? 1.0 : ((lane > j1 && lane < n) ? x * sc : 0.0);
        if (lane == k) { scv = sc; tauv = tk; }
        ksteps = k + 1;
        if (lane == 0) { e[k] = beta; tau[k] = tk; }
        if (!unclean && fabs(beta) <= tol_c) {
            // the Krylov space is exhausted: everything behind row k must be the null space
            double m = 0.0;
#pragma unroll
            for (int i = 0; i < N; ++i)
                if (i > k && lane > k) m = fmax(m, fabs(a[i]));
            m = wave_max(m);
            if (m <= tol_n) {
                if (lane == 0) tau[k] = 0.0;
                if (lane == k) tauv = 0.0;
                coupling = beta;
                kk = k + 1;
                const double xt = row_of<N>(a, j1);   // first diagonal entry of the null block (eigensolver path)
                if (lane == j1) d[j1] = xt;
                break;
            }
            // not null: no clean split.  The tridiagonalisation is completed (plain Householder steps from here) for
            // the in-wave eigensolver; the return value says so.
            unclean = true;
        }
        if (tk != 0.0) two_sided(k, tk, v);
    }
    if (kk == n) {   // ran to the end: the last diagonal entry
        const double x = row_of<N>(a, n - 1);
        if (lane == n - 1) d[n - 1] = x;
    }
    if (lane == 0) out4[3] = coupling;
    MVMC_WAVE_SYNC();
    return unclean ? -1 : kk;
}

// The reflectors, packed two rows to a register: row z only matters on the lanes >= z + 2, so the payload of row N - 3 - z (lanes
// N - 1 - z .. N - 1, z + 1 values) moves into the lanes 0 .. z of row z.  (N - 2) / 2 registers instead of N: what has to stay
// live through the trust-region solve, whose cyclic reduction needs ~70 registers of its own.
template <int N>
__device__ __forceinline__ void pack_reflectors(const double (&a)[N], double (&pk)[(N - 2) / 2]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int z = 0; z < (N - 2) / 2; ++z) {
        const int zp = N - 3 - z;
        const double hi = __shfl(a[zp], lane + zp + 2, 64);
        pk[z] = lane >= z + 2 ? a[z] : hi;
    }
}

// Q c from the packed reflectors (pack_reflectors; scv, tauv: tridiag_krylov_w1's per-lane scalars): reflectors kk-2 .. 0, then
// the first one.  Same operations, in the same order, as apply_q_w1 on the dumped vectors.
template <int N>
__device__ __forceinline__ double apply_q_packed(const double (&pk)[(N - 2) / 2], double scv, double tauv, const double* v0, double tau0,
                                                 int kk, int n, double cj) {
    constexpr int H = (N - 2) / 2;
    const int lane = threadIdx.x & 63;
    auto wave_sum_dpp = [](double v) { return wave_sum_rows<(N + 15) / 16>(v); };   // (v vanishes on the lanes >= n; same bits)
    auto lane_value = [](double v, int src) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
    };
#pragma unroll
    for (int z = N - 2; z >= 0; --z)
        if (z <= kk - 2) {
            const double sc = lane_value(scv, z), tk = lane_value(tauv, z);
            double x = 0.0;
            if (z < H) x = pk[z];
            else if (z < N - 2) x = __shfl(pk[N - 3 - z], lane - (z + 2), 64);
            const double v = lane == z + 1 ? 1.0 : ((lane > z + 1 && lane < n) ? x * sc : 0.0);
            cj -= tk * wave_sum_dpp(v * cj) * v;
        }
    if (tau0 != 0.0) {
        const double v = lane < n ? v0[lane] : 0.0;
        cj -= tau0 * wave_sum_dpp(v * cj) * v;
    }
    return cj;
}

// The reflectors of the first `ksteps` steps to global memory, in apply_q_w1's layout (eigenbasis fallback only).
template <int N>
__device__ __forceinline__ void dump_reflectors(const double (&pk)[(N - 2) / 2], double scv, int ksteps, int n, mvmc_gdouble* __restrict__ hh) {
    constexpr int H = (N - 2) / 2;
    const int lane = threadIdx.x & 63;
    auto lane_value = [](double v, int src) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
    };
#pragma unroll
    for (int z = 0; z < N - 1; ++z)
        if (z < ksteps) {
            const double sc = lane_value(scv, z);
            double x = 0.0;
            if (z < H) x = pk[z];
            else if (z < N - 2) x = __shfl(pk[N - 3 - z], lane - (z + 2), 64);
            const double v = lane == z + 1 ? 1.0 : ((lane > z + 1 && lane < n) ? x * sc : 0.0);
            if (lane > z && lane < n) hh[z * 64 + lane] = v;
        }
}

// Q c for tridiag_krylov_w1 (one wave; lane j holds component j): reflectors kk-2 .. 0 from hh, then the first one.
__device__ inline double apply_q_w1(const mvmc_gdouble* __restrict__ hh, const double* tau, const double* v0, double tau0, int kk,
                                    int n, double cj) {
    const int lane = threadIdx.x & 63;
    int k = kk - 2;
    auto row = [&](int r) { return (lane > r && lane < n) ? hh[r * 64 + lane] : 0.0; };
    for (; k >= 3; k -= 4) {   // four rows in flight: the loads do not depend on the running vector
        const double va = row(k), vb = row(k - 1), vc = row(k - 2), vd = row(k - 3);
        const double ta = tau[k], tb = tau[k - 1], tc = tau[k - 2], td = tau[k - 3];
        cj -= ta * wave_sum_dpp(va * cj) * va;
        cj -= tb * wave_sum_dpp(vb * cj) * vb;
        cj -= tc * wave_sum_dpp(vc * cj) * vc;
        cj -= td * wave_sum_dpp(vd * cj) * vd;
    }
    for (; k >= 0; --k) {
        const double v = row(k);
        cj -= tau[k] * wave_sum_dpp(v * cj) * v;
    }
    if (tau0 != 0.0) {
        const double v = lane < n ? v0[lane] : 0.0;
        cj -= tau0 * wave_sum_dpp(v * cj) * v;
    }
    return cj;
}


// ---------------------------------------------------------------------------------------------------------------
// Eigendecomposition of the m x m tridiagonal T (d, e) on one wave -- the single-wave form of steps 2-3b of eigh():
// lane i owns eigenpair i.  Eigenvalues by bisection on Sturm counts (60 halvings of the Gershgorin interval),
// eigenvectors by twisted factorisation with the forward pivots in registers and the backward pivots parked in
// global memory (Zg, which then receives the vectors), runs of close eigenvalues (gap < 1e-7 lam_max)
// re-orthogonalised by classical Gram-Schmidt through v_readlane.  The numerically-null cluster
// (lam <= 1e-13 lam_max) is not resolved: zero vectors, zero eigenvalues, as in eigh().
//   lam[0..64) (LDS, ascending; 0 beyond m), Zg[j * 64 + i] = component j of eigenvector i (global, m rows),
//   dsc, e2sc, wsh: 64 doubles of LDS scratch each.  Returns the size k0 of the null cluster.
// No back-transformation: the caller works in the T basis (Q^T g = beta0 e_1 there) and applies Q to the one vector
// it needs (apply_q_w1).
// ---------------------------------------------------------------------------------------------------------------
template <int N>
__device__ __noinline__ int tri_eigh_w1(const double* d, const double* e, int m, double* lam, mvmc_gdouble* __restrict__ Zg,
                                        double* dsc, double* e2sc, double* wsh) {
    MVMC_ASSUME_LDS(d); MVMC_ASSUME_LDS(e); MVMC_ASSUME_LDS(lam); MVMC_ASSUME_LDS(dsc); MVMC_ASSUME_LDS(e2sc); MVMC_ASSUME_LDS(wsh);
    const int lane = threadIdx.x & 63;
    auto lane_value = [](double v, int src) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
    };
    // ---- eigenvalues ----
    double lo, hi, tscale;
    {
        double gl = 1e300, gu = -1e300;
        if (lane < m) {
            const double r = (lane > 0 ? fabs(e[lane - 1]) : 0.0) + (lane < m - 1 ? fabs(e[lane]) : 0.0);
            gl = d[lane] - r; gu = d[lane] + r;
        }
        for (int off = 32; off > 0; off >>= 1) { gl = fmin(gl, __shfl_xor(gl, off, 64)); gu = fmax(gu, __shfl_xor(gu, off, 64)); }
        const double bn = fmax(fabs(gl), fabs(gu));
        tscale = bn > 0.0 ? bn : 1.0;
        lo = (gl - 2.2e-14 * bn) / tscale - 1e-290; hi = (gu + 2.2e-14 * bn) / tscale + 1e-290;
    }
    dsc[lane] = lane < m ? d[lane] / tscale : 4.0;
    {
        const double es = lane < m - 1 ? e[lane] / tscale : 0.0;
        e2sc[lane] = es * es;
    }
    MVMC_WAVE_SYNC();
    const int nblk = (m - 1 + 7) >> 3;
    for (int round = 0; round < 60; ++round) {
        const double mid = 0.5 * (lo + hi);
        const int c = sturm_count(dsc, e2sc, nblk, mid);
        if (c > lane) hi = mid; else lo = mid;
    }
    const double lam_i = lane < m ? 0.5 * (lo + hi) * tscale : 0.0;
    MVMC_WAVE_SYNC();
    lam[lane] = lam_i;
    e2sc[lane] = lane < m - 1 ? e[lane] * e[lane] : 0.0;   // unscaled squares for the twisted factorisation
    MVMC_WAVE_SYNC();
    const double lmax = fmax(fabs(lam[m - 1]), fabs(lam[0]));
    const double tol0 = 1e-13 * lmax;
    int k0 = 0;
    for (int i = 0; i < m; ++i) k0 += lam[i] <= tol0;  // ascending: the null cluster is lam[0..k0)
    const double pivmin = 1e-16 * lmax + 1e-300;
    // shifts: eigenvalues closer than a few ulps of lam_max are pushed apart (as LAPACK's dstein does)
    if (lane == 0) {
        const double sep = 4.4e-16 * lmax;
        double prev = -1e300;
        for (int i = 0; i < m; ++i) { prev = fmax(lam[i], prev + sep); wsh[i] = prev; }
    }
    MVMC_WAVE_SYNC();
    const bool on = lane < m && lane >= k0;
    const double l = on ? wsh[lane] : 0.0;
    auto clamp = [&](double p) { return fabs(p) < pivmin ? (p < 0.0 ? -pivmin : pivmin) : p; };
    // ---- eigenvectors: forward pivots D+ in z[], backward pivots D- through Zg ----
    double z[N];
    {
        double dp = d[0] - l;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (j < m - 1) {
                dp = clamp(dp);
                z[j] = dp;
                dp = (d[j + 1] - l) - e2sc[j] * fast_rcp64(dp);
            } else if (j == m - 1) {
                z[j] = dp;
            } else {
                z[j] = 0.0;
            }
        }
    }
    int r = 0;
    {
        double dm = d[m - 1] - l, gmin = 1e300;
#pragma unroll
        for (int j = N - 1; j >= 0; --j) {
            if (j < m) {
                if (j > 0) dm = clamp(dm);
                Zg[j * 64 + lane] = dm;
                const double gam = fabs(z[j] + dm - (d[j] - l));
                if (gam <= gmin) { gmin = gam; r = j; }
                if (j > 0) dm = (d[j - 1] - l) - e2sc[j - 1] * fast_rcp64(dm);
            }
        }
    }
    // z_r = 1; downward with L_j = e_j / D+_j (in place: z[j] still holds D+_j when it is consumed)
    {
        double zz = 1.0;
#pragma unroll
        for (int j = N - 2; j >= 0; --j) {
            if (j < m - 1) {
                if (j + 1 == r) zz = 1.0;
                if (j < r) { zz = -e[j] * zz * fast_rcp64(clamp(z[j])); z[j] = zz; }
            }
        }
    }
    // upward with U_j = e_j / D-_{j+1}; the pivots come back from global memory eight rows at a time
    {
        double zz = 1.0;
#pragma unroll
        for (int c = 0; c < N; c += 8) {
            if (c < m - 1) {
                double dmv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) dmv[u] = (c + u + 1 < m && c + u + 1 < N) ? Zg[(c + u + 1) * 64 + lane] : 1.0;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = c + u;
                    if (j + 1 < N && j < m - 1) {
                        if (j == r) zz = 1.0;
                        if (j >= r) { zz = -e[j] * zz * fast_rcp64(clamp(dmv[u])); z[j + 1] = zz; }
                    }
                }
            }
        }
    }
    {
        double nrm = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (j == r) z[j] = 1.0;
            if (j >= m) z[j] = 0.0;
            nrm += z[j] * z[j];
        }
        const double inv = 1.0 / sqrt(nrm);
#pragma unroll
        for (int j = 0; j < N; ++j) z[j] = on ? z[j] * inv : 0.0;
    }
    // ---- runs of close eigenvalues: classical Gram-Schmidt of vector i against the earlier vectors of its run ----
    {
        const double gtol = 1e-7 * lmax;
        int start = k0;
        for (int i = k0 + 1; i < m; ++i) {   // uniform control flow: lam is shared
            if (lam[i] - lam[i - 1] >= gtol) { start = i; continue; }
            for (int q = start; q < i; ++q) {
                double sd = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) sd += lane_value(z[j], q) * z[j];
                if (lane == i) wsh[q - start] = sd;   // dots against the unmodified vector i
            }
            MVMC_WAVE_SYNC();
            for (int q = start; q < i; ++q) {
                const double sq = wsh[q - start];
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const double zq = lane_value(z[j], q);
                    if (lane == i) z[j] -= sq * zq;
                }
            }
            double nn = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) nn += z[j] * z[j];
            if (lane == i) {
                const double inv = 1.0 / sqrt(nn);
#pragma unroll
                for (int j = 0; j < N; ++j) z[j] *= inv;
            }
            MVMC_WAVE_SYNC();
        }
    }
#pragma unroll
    for (int j = 0; j < N; ++j)
        if (j < m) Zg[j * 64 + lane] = z[j];
    MVMC_WAVE_SYNC();
    if (lane < k0) lam[lane] = 0.0;
    MVMC_WAVE_SYNC();
    return k0;
}

// solve_lsq_trust_region (common.py:57-168) in the eigenbasis (lam, suf; one lane per eigenpair, m < 64), the
// rank-deficient branch with the virtual absorber at lane m (the absorber).  cv <- coefficients.
__device__ inline double tr_solve_eig_w1(const double* lamv, const double* sufv, int m, double Delta, double alpha0, double gg,
                                         double* cv, double* pred, double* pnorm) {
    const int lane = threadIdx.x & 63;
    const bool on = lane <= m;
    double lam = lane < m ? lamv[lane] : 1.0, suf = lane < m ? sufv[lane] : 0.0;
    if (lane == m) { lam = 0.0; suf = 1e-8 * sqrt(gg); }
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
    if (lane < m) cv[lane] = c;
    *pred = -(0.5 * wave_sum(lam * c * c) + wave_sum(suf * c));
    *pnorm = sqrt(wave_sum(c * c));  // |step_h| including the absorber (== Delta up to rounding, as in SciPy)
    return alpha;
}

// y = Z c: component j to lane j (Zg as written by tri_eigh_w1, c one coefficient per lane, zero beyond m)
__device__ inline double eig_combine_w1(const mvmc_gdouble* __restrict__ Zg, int m, double c) {
    const int lane = threadIdx.x & 63;
    double y = 0.0;
    int j = 0;
    for (; j + 3 < m; j += 4) {
        const double z0 = Zg[j * 64 + lane], z1 = Zg[(j + 1) * 64 + lane], z2 = Zg[(j + 2) * 64 + lane], z3 = Zg[(j + 3) * 64 + lane];
        const double t0 = wave_sum_dpp(z0 * c), t1 = wave_sum_dpp(z1 * c), t2 = wave_sum_dpp(z2 * c), t3 = wave_sum_dpp(z3 * c);
        y = lane == j ? t0 : (lane == j + 1 ? t1 : (lane == j + 2 ? t2 : (lane == j + 3 ? t3 : y)));
    }
    for (; j < m; ++j) {
        const double t = wave_sum_dpp(Zg[j * 64 + lane] * c);
        if (lane == j) y = t;
    }
    return y;
}

}  // namespace eightri
