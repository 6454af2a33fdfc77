// TWO stage-1 trust-region models on ONE wave (included by mvmc_ik1.hip inside its anonymous namespace; SMALL layout of the chain kernel).
//
// The stage-1 model of a solve (ik1_model_step_r<30, 0>: 30 reduced columns) uses 30 of the wave's 64 lanes, and most of what it
// issues is per-step overhead that costs the same with 30 lanes as with 64 (a Householder step is ~110 fixed vector instructions --
// two wave reductions, the reflector's scalars -- beside ~50 fused multiply-adds).  The chain kernel solves the people of a frame side
// by side, one wave each, and the waves reach their model builds at about the same time: here the waves of a pair (0, 1), (2, 3) meet
// at that point, and whoever arrives SECOND builds both models -- its own in the lanes 0 .. 31, the partner's in the lanes 32 .. 63 -- while
// the partner sleeps.  Same instructions, two models: the stage-1 models were 25 % of a step's time (doubling them: 18.2 -> 22.75 ms).
//
// Bit-identical to the one-model form by construction: lane 32 h + l of the pair does what lane l of the single model does, in the same
// order; a reduction over a half is the single form's reduction over the wave (whose lanes 32 .. 63 hold zeros: the row totals are added
// in the same order); the parallel cyclic reduction of the trust-region solve drops its sixth round, which is an exact no-op below 33
// rows.  Only the common path is paired: a model that would leave it in EITHER half (gimbal lock, the gradient test, a sub-diagonal
// that collapses, a leading block that fails its checks) is abandoned, and both waves build their models the ordinary way -- the model is
// a deterministic function of the solve's LDS state, which the pair function only reads.
//
// The meeting (ik1_pair_sync) uses two LDS words per pair, both inside the solve blocks (Ik1Shared::pairw): the even wave's is the
// MAILBOX (0 = empty, a posted request, or PAIR_TAKEN), the odd wave's the COMPLETION word (a counter in the low half, an "away" bit per
// wave above it: a wave that is not inside a stage-1 solve will not come to a meeting, so nobody waits for it).
//   first arriver:   CAS(mailbox, 0 -> request); then waits until the request is taken (-> waits for the counter to move, reads its
//                    results from its own solve block) or the partner is away (CAS(mailbox, request -> 0): withdrawn -> builds alone;
//                    if that fails the request was taken in the meantime).
//   second arriver:  finds the partner's request, CAS(mailbox, request -> PAIR_TAKEN), builds both models, empties the mailbox, bumps
//                    the counter.
// Every wait ends: a waiting wave's partner either arrives at a meeting, or leaves stage 1 (raises its away bit), or finishes the frame's
// solves (chain_ik raises the bit for it) -- none of which depends on the waiting wave; the builder of a pair never waits.
#pragma once

namespace pairw {

constexpr int PAIR_TAKEN = 0x40000000;
constexpr int PAIR_ABORT = -2;     // the pair function's result: not the common path in one of the halves
constexpr int PAIR_ALONE = -1;     // ik1_pair_sync: no partner, build the ordinary way

// ---- half-wave forms of the wave helpers (results per half: lanes 0 .. 31, lanes 32 .. 63) ----
__device__ __forceinline__ double hsum(double v) {
    v += dpp_mov<0x128>(v);  // row_ror:8
    v += dpp_mov<0x124>(v);  // row_ror:4
    v += dpp_mov<0x122>(v);  // row_ror:2
    v += dpp_mov<0x121>(v);  // row_ror:1
    const int lo = __double2loint(v), hi = __double2hiint(v);
    double ta = 0.0, tb = 0.0;   // (wave_sum_dpp: t = 0 + r0 + r1 + r2 + r3, in this order)
    ta += __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    ta += __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    tb += __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    tb += __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return (threadIdx.x & 32) ? tb : ta;
}
__device__ __forceinline__ double hmax(double v) {
    v = fmax(v, dpp_mov<0x128>(v));
    v = fmax(v, dpp_mov<0x124>(v));
    v = fmax(v, dpp_mov<0x122>(v));
    v = fmax(v, dpp_mov<0x121>(v));
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double ta = fmax(__hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0)),
                           __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16)));
    const double tb = fmax(__hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32)),
                           __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48)));
    return (threadIdx.x & 32) ? tb : ta;
}
// v of lane `src` (wave-uniform, < 32) of the lane's own half
__device__ __forceinline__ double hval(double v, int src) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double a = __hiloint2double(__builtin_amdgcn_readlane(hi, src), __builtin_amdgcn_readlane(lo, src));
    const double b = __hiloint2double(__builtin_amdgcn_readlane(hi, src + 32), __builtin_amdgcn_readlane(lo, src + 32));
    return (threadIdx.x & 32) ? b : a;
}
__device__ __forceinline__ bool any_lane(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0ull; }

// ---- eightri::tridiag_krylov_w1<N> for two matrices (N <= 32): lane 32 h + l owns column l of matrix h.  d, e, tau, v0, vb, pb, out4
// are the lane's half's vectors.  Returns false if a sub-diagonal collapses in either half (the single form's stopping rules are not
// restated here: the caller abandons the pair).  On success the tridiagonalisation is complete: kk = n in both halves. ----
template <int N>
__device__ __forceinline__ bool tridiag_krylov_pair(double (&a)[N], double gj, int n, double* d, double* e, double* tau, double* v0,
                                                    double* vb, double* pb, double* out4, double& scv, double& tauv) {
    static_assert(N % 2 == 0 && N <= 32, "two matrices of at most 32 columns");
    const int l = threadIdx.x & 31;
    auto reflector = [&](double alpha, double sig, double& tk, double& beta, double& sc) {   // (eightri::tridiag_krylov_w1's, verbatim)
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
#ifdef MVMC_TRI_HB
    constexpr int HB = MVMC_TRI_HB;
#else
    constexpr int HB = N <= 40 ? 5 : 3;
#endif
#ifdef MVMC_TRI_CHB
    constexpr int CH = 10, CHB = MVMC_TRI_CHB;
#else
    constexpr int CH = 10, CHB = 10;
#endif
    static_assert(N % CH == 0 && N % CHB == 0, "rows per chunk");
    auto rank2 = [&](int k, double vj, double wj) {
#pragma unroll
        for (int c = 0; c < N; c += CH)
            if (c + CH - 1 > k) {
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
        MVMC_WAVE_SYNC();
    };
    auto two_sided = [&](int k, double tk, double vj) {
        vb[l] = vj;
        MVMC_WAVE_SYNC();
        double p0 = 0.0, p1 = 0.0;
#pragma unroll
        for (int c = 0; c < N; c += CHB)
            if (c + CHB - 1 > k) {
#ifdef MVMC_TRI_PB
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
        const double p = tk * (p0 + p1);
        const double h = 0.5 * tk * hsum(p * vj);
        const double wj = l > k ? p - h * vj : 0.0;
        pb[l] = wj;
        MVMC_WAVE_SYNC();
        rank2(k, vj, wj);
    };
    double anorm;
    {
        double cs = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) cs += fabs(a[i]);
        anorm = hmax(cs);
    }
    const double tol_c = 1e-8 * anorm;
    {   // first reflector: H_g g = beta0 e_1
        const double x = l < n ? gj : 0.0;
        const double alpha = hval(x, 0);
        const double sig = hsum(l > 0 ? x * x : 0.0);
        double tk, beta, sc;
        reflector(alpha, sig, tk, beta, sc);
        const double v = l == 0 ? 1.0 : x * sc;
        v0[l] = v;
        if (l == 0) { out4[0] = beta; out4[1] = tk; out4[2] = anorm; }
        if (any_lane(tk != 0.0)) two_sided(-1, tk, v);   // (a half with tk = 0: p = w = 0, its matrix keeps its values)
    }
    scv = 0.0; tauv = 0.0;
    for (int k = 0; k < n - 1; ++k) {
        const int j1 = k + 1;
        const double x = eightri::row_of<N>(a, k);
        if (l == k) d[k] = x;
        const double alpha = hval(x, j1);
        const double sig = hsum((l > j1 && l < n) ? x * x : 0.0);
        double tk, beta, sc;
        reflector(alpha, sig, tk, beta, sc);
        const double v = l == j1 ? 1.0 : ((l > j1 && l < n) ? x * sc : 0.0);
        if (l == k) { scv = sc; tauv = tk; }
        if (l == 0) { e[k] = beta; tau[k] = tk; }
        if (any_lane(fabs(beta) <= tol_c)) return false;   // the Krylov space of one half is exhausted: the single form decides what that means
        if (any_lane(tk != 0.0)) two_sided(k, tk, v);
    }
    {
        const double x = eightri::row_of<N>(a, n - 1);
        if (l == n - 1) d[n - 1] = x;
    }
    if (l == 0) out4[3] = 0.0;   // coupling: no collapse
    MVMC_WAVE_SYNC();
    return true;
}

// eightri::pack_reflectors for two matrices
template <int N>
__device__ __forceinline__ void pack_reflectors_pair(const double (&a)[N], double (&pk)[(N - 2) / 2]) {
    const int lane = threadIdx.x & 63, l = lane & 31;
#pragma unroll
    for (int z = 0; z < (N - 2) / 2; ++z) {
        const int zp = N - 3 - z;
        const double hi = __shfl(a[zp], lane + zp + 2, 64);   // (lanes l <= z + 1 of the half: source l + zp + 2 <= N - 1, inside the half)
        pk[z] = l >= z + 2 ? a[z] : hi;
    }
}

// eightri::apply_q_packed for two matrices, kk = n (the complete tridiagonalisation); tau0 per half
template <int N>
__device__ __forceinline__ double apply_q_packed_pair(const double (&pk)[(N - 2) / 2], double scv, double tauv, const double* v0, double tau0,
                                                      int n, double cj) {
    constexpr int H = (N - 2) / 2;
    const int lane = threadIdx.x & 63, l = lane & 31;
#pragma unroll
    for (int z = N - 2; z >= 0; --z)
        if (z <= n - 2) {
            const double sc = hval(scv, z), tk = hval(tauv, z);
            double x = 0.0;
            if (z < H) x = pk[z];
            else if (z < N - 2) x = __shfl(pk[N - 3 - z], lane - (z + 2), 64);
            const double v = l == z + 1 ? 1.0 : ((l > z + 1 && l < n) ? x * sc : 0.0);
            cj -= tk * hsum(v * cj) * v;
        }
    {
        const double v = l < n ? v0[l] : 0.0;
        const double cn = cj - tau0 * hsum(v * cj) * v;
        cj = tau0 != 0.0 ? cn : cj;
    }
    return cj;
}

// eightri::dump_reflectors for the lanes of the halves named by `on` (ksteps = n - 1): hh is the lane's half's scratch
template <int N>
__device__ __forceinline__ void dump_reflectors_pair(const double (&pk)[(N - 2) / 2], double scv, int n, mvmc_gdouble* hh, bool on) {
    constexpr int H = (N - 2) / 2;
    const int lane = threadIdx.x & 63, l = lane & 31;
#pragma unroll
    for (int z = 0; z < N - 1; ++z)
        if (z < n - 1) {
            const double sc = hval(scv, z);
            double x = 0.0;
            if (z < H) x = pk[z];
            else if (z < N - 2) x = __shfl(pk[N - 3 - z], lane - (z + 2), 64);
            const double v = l == z + 1 ? 1.0 : ((l > z + 1 && l < n) ? x * sc : 0.0);
            if (on && l > z && l < n) hh[z * 64 + l] = v;
        }
}

// eightri::tr_solve_tri<false> for two tridiagonal matrices of n <= 32 rows (d, e, rh, cout: the lane's half's vectors; Delta, alpha0,
// gg, pivmin per half).  Five rounds of the cyclic reduction: the sixth (distance 32) is an exact no-op below 33 rows -- after round
// t the rows j < 2^t have no lower neighbour left and the rows j + 2^t >= n no upper one, so both multipliers are zeros.  A half
// whose Newton iteration on alpha has ended keeps its alpha while the other half finishes.
__device__ __forceinline__ double tr_solve_tri_pair(const double* d, const double* e, const double* rh, int n, double Delta, double alpha0,
                                                    double gg, double pivmin, double* cout, double* pred, double* pnorm) {
    const int l = threadIdx.x & 31;
    const bool on = l < n;
    const double a2 = 1e-16 * gg;
    const double dj = on ? d[l] : 1.0, el = (on && l > 0) ? e[l - 1] : 0.0, eu = (l < n - 1) ? e[l] : 0.0;
    const double rj = on ? rh[l] : 0.0;
    double yj, yy, yw, ry, yz;
    auto evaluate = [&](double alpha, bool want_z) {
        double a = el, bq = on ? dj + alpha : 1.0, c = eu, r = rj;
        double k1[5], k2[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const int sft = 1 << t;
            const bool lo_ok = l >= sft, hi_ok = l + sft < 32;
            double am = __shfl_up(a, sft, 64), bm = __shfl_up(bq, sft, 64), cm = __shfl_up(c, sft, 64), rm = __shfl_up(r, sft, 64);
            double ap = __shfl_down(a, sft, 64), bp = __shfl_down(bq, sft, 64), cp = __shfl_down(c, sft, 64), rp = __shfl_down(r, sft, 64);
            am = lo_ok ? am : 0.0; bm = lo_ok ? bm : 1.0; cm = lo_ok ? cm : 0.0; rm = lo_ok ? rm : 0.0;
            ap = hi_ok ? ap : 0.0; bp = hi_ok ? bp : 1.0; cp = hi_ok ? cp : 0.0; rp = hi_ok ? rp : 0.0;
            k1[t] = a * fast_rcp64(bm);
            k2[t] = c * fast_rcp64(bp);
            bq = bq - cm * k1[t] - ap * k2[t];
            r = r - rm * k1[t] - rp * k2[t];
            a = -am * k1[t];
            c = -cp * k2[t];
        }
        if (bq < pivmin) bq = pivmin;
        const double binv = fast_rcp64(bq);
        yj = on ? r * binv : 0.0;
        double ym = __shfl_up(yj, 1, 64), yp = __shfl_down(yj, 1, 64);
        ym = l > 0 ? ym : 0.0; yp = l < 31 ? yp : 0.0;
        const double wj = on ? el * ym + dj * yj + eu * yp : 0.0;  // (T y)_j
        yy = hsum(yj * yj);
        yw = hsum(yj * wj);
        ry = hsum(rj * yj);
        if (want_z) {
            double rz = yj;
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                const int sft = 1 << t;
                double rm = __shfl_up(rz, sft, 64), rp = __shfl_down(rz, sft, 64);
                rm = l >= sft ? rm : 0.0; rp = l + sft < 32 ? rp : 0.0;
                rz = rz - rm * k1[t] - rp * k2[t];
            }
            const double zj = on ? rz * binv : 0.0;
            yz = hsum(yj * zj);
        }
    };
    double alpha_upper = sqrt(gg + a2) / Delta;
    double alpha_lower = 0.0;
    double alpha = (alpha0 == 0.0) ? fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper)) : alpha0;
    bool active = true;
    for (int it = 0; it < 10; ++it) {
        if (!any_lane(active)) break;
        if (active && (alpha < alpha_lower || alpha > alpha_upper))
            alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
        evaluate(alpha, true);
        const double ia = 1.0 / alpha;
        const double s1 = yy + a2 * ia * ia;
        const double s3 = yz + a2 * ia * ia * ia;
        const double p_norm = sqrt(s1);
        const double phi = p_norm - Delta;
        const double phi_prime = -s3 / p_norm;
        const double ratio = phi / phi_prime;
        if (active) {
            if (phi < 0) alpha_upper = alpha;
            alpha_lower = fmax(alpha_lower, alpha - ratio);
            alpha -= (phi + Delta) * ratio / Delta;
            if (fabs(phi) < 0.01 * Delta) active = false;
        }
    }
    evaluate(alpha, false);
    const double ia = 1.0 / alpha;
    const double pn = sqrt(yy + a2 * ia * ia);
    const double sc = Delta / pn;
    if (on) cout[l] = -yj * sc;
    const double lcc = sc * sc * yw;
    const double sfc = -sc * ry - a2 * ia * sc;
    *pred = -(0.5 * lcc + sfc);
    *pnorm = sc * pn;
    return alpha;
}

// eightri::krylov_block_ok for two complete tridiagonalisations (kk = n): the Sturm count of each half's matrix at 1e-13 |M|_inf.
__device__ __forceinline__ bool krylov_block_ok_pair(const double* d, const double* e, int n, double anorm, double* dsc, double* e2sc) {
    const int l = threadIdx.x & 31;
    const double ts = anorm > 0.0 ? anorm : 1.0;
    dsc[l] = l < n ? d[l] / ts : 4.0;
    dsc[32 + l] = 4.0;                       // (the count reads whole blocks of eight: neutral steps behind the matrix)
    const double es = l < n - 1 ? e[l] / ts : 0.0;
    e2sc[l] = es * es;
    e2sc[32 + l] = 0.0;
    MVMC_WAVE_SYNC();
    return !any_lane(eightri::sturm_count(dsc, e2sc, (n - 1 + 7) >> 3, 1e-13) != 0);
}

}  // namespace pairw

// ---------------------------------------------------------------------------------------------
// ik1_model_step_r<30, 0> (budget left: the model AND its first trial) for the solve blocks S0 (lanes 0 .. 31) and S1 = S0 + dS bytes
// (lanes 32 .. 63).  Inputs per solve: the FK state and blocks of the last evaluation, Delta in sc[8], alpha in sc[2]; dumpbits: bit h =
// the reflectors of half h go to its scratch (hh0, hh0 + dhh doubles).  Outputs per solve: exactly what the single function leaves
// (sc[0 .. 9], sc[11], the solver vectors, the trial point in xn).  *code_out = 1 (trial made in both halves) or PAIR_ABORT.
// ---------------------------------------------------------------------------------------------
__device__ __noinline__ void ik1_model_pair_r(Ik1Shared& S0in, int dS, const Ik1Tables& T, double gtol, mvmc_gdouble* __restrict__ hh0,
                                              int dhh, int dumpbits, int* code_out) {
    MVMC_ASSUME_LDS(&S0in);
    MVMC_ASSUME_LDS(&T);
    using namespace arrow;
    constexpr int N = Dim<0>::NR, na = N;
    static_assert(N == 30, "the stage-1 model");
    Ik1Shared& S0 = *uni(&S0in);
    dS = uni(dS); dhh = uni(dhh);
    Ik1Shared& S1 = *reinterpret_cast<Ik1Shared*>(reinterpret_cast<char*>(&S0) + dS);
    MVMC_ASSUME_LDS(&S1);
    hh0 = uni(hh0);
    const int lane = threadIdx.x & 63, l = lane & 31;
    const bool hi = lane >= 32;
    Ik1Shared& S = *reinterpret_cast<Ik1Shared*>(reinterpret_cast<char*>(&S0) + (hi ? dS : 0));   // the lane's solve
    MVMC_ASSUME_LDS(&S);
    const int nae = uni(T.na[0]);
    const bool on = l < na;
    const int kind = on ? Ik1Tables::rkind(l) : 4;
    const double Delta = S.sc[8], alpha_in = S.sc[2];
    double* refl = S.xn;
    bool bad = false;
    if (l < 5) bad = structure_reflectors(S.pos, S.hs, S.Rg, T, l, refl + l * 11);
    if (pairw::any_lane(bad)) { *code_out = pairw::PAIR_ABORT; return; }
    MVMC_WAVE_SYNC();
    int ja = 0, jb = 0;
    double wa0 = 0.0, wa1 = 0.0, wa2 = 0.0, wb0 = 0.0, wb1 = 0.0, wb2 = 0.0;
    if (kind == 1) {
        ja = jb = T.rja[l];
        const int jc = T.rjc[l];
        const double b[3] = {jc == 0 ? 1.0 : 0.0, jc == 1 ? 1.0 : 0.0, jc == 2 ? 1.0 : 0.0};
        double w[3];
        omega_of(&S.hs[ja * 4], &S.Rg[(ja ? T.parents[ja] : 0) * 9], ja == 0, b, w);
        wa0 = w[0]; wa1 = w[1]; wa2 = w[2];
    } else if (kind == 3) {
        const int s = T.rja[l], bc = T.rjc[l];
        ja = T.s_ja[s]; jb = T.s_jb[s];
        double z[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) z[i] = bc == i ? 1.0 : 0.0;
        apply_reflectors(refl + s * 11, z);
        double w[3];
        omega_of(&S.hs[ja * 4], &S.Rg[T.parents[ja] * 9], false, z, w);
        wa0 = w[0]; wa1 = w[1]; wa2 = w[2];
        omega_of(&S.hs[jb * 4], &S.Rg[T.parents[jb] * 9], false, z + 3, w);
        wb0 = w[0]; wb1 = w[1]; wb2 = w[2];
    }
    const int jcc = (kind == 0) ? (int)T.rjc[l] : 0;
#ifdef MVMC_IK_PARK
    double* wl = S.sv;
    wl[l] = wa0; wl[64 + l] = wa1; wl[128 + l] = wa2; wl[192 + l] = wb0; wl[256 + l] = wb1; S.tmp[192 + l] = wb2;
    MVMC_WAVE_SYNC();
#endif
    double a[N];
#pragma unroll
    for (int i = 0; i < N; ++i) a[i] = 0.0;
    double gj = 0.0;
    double* db = S.tmp;
    for (int k = 0; k < NOBS; ++k) {
        const int K = kIkSkel[k];
        double d0 = 0.0, d1 = 0.0, d2 = 0.0;
        if (kind == 0) {
            d0 = jcc == 0 ? 1.0 : 0.0; d1 = jcc == 1 ? 1.0 : 0.0; d2 = jcc == 2 ? 1.0 : 0.0;
        } else if (kind == 1 || kind == 3) {
            const int anc = T.anc[K];
#ifdef MVMC_IK_PARK
            const double wa0 = wl[l], wa1 = wl[64 + l], wa2 = wl[128 + l];
#endif
            if ((anc >> ja) & 1) {
                const double r0 = S.pos[K * 3] - S.pos[ja * 3], r1 = S.pos[K * 3 + 1] - S.pos[ja * 3 + 1], r2 = S.pos[K * 3 + 2] - S.pos[ja * 3 + 2];
                d0 = wa1 * r2 - wa2 * r1; d1 = wa2 * r0 - wa0 * r2; d2 = wa0 * r1 - wa1 * r0;
            }
            if (kind == 3 && ((anc >> jb) & 1)) {
#ifdef MVMC_IK_PARK
                const double wb0 = wl[192 + l], wb1 = wl[256 + l], wb2 = S.tmp[192 + l];
#endif
                const double r0 = S.pos[K * 3] - S.pos[jb * 3], r1 = S.pos[K * 3 + 1] - S.pos[jb * 3 + 1], r2 = S.pos[K * 3 + 2] - S.pos[jb * 3 + 2];
                d0 += wb1 * r2 - wb2 * r1; d1 += wb2 * r0 - wb0 * r2; d2 += wb0 * r1 - wb1 * r0;
            }
        }
        const double* W = &S.Wk[k * 6];
        const double y0 = W[0] * d0 + W[1] * d1 + W[2] * d2;
        const double y1 = W[1] * d0 + W[3] * d1 + W[4] * d2;
        const double y2 = W[2] * d0 + W[4] * d1 + W[5] * d2;
        gj += d0 * S.tk[k * 3] + d1 * S.tk[k * 3 + 1] + d2 * S.tk[k * 3 + 2];
        MVMC_WAVE_SYNC();
        db[l * 3] = d0; db[l * 3 + 1] = d1; db[l * 3 + 2] = d2;
        MVMC_WAVE_SYNC();
#ifdef MVMC_IK_GR
        constexpr int GR = MVMC_IK_GR, GL = GR * 3 / 2;
#else
        constexpr int GR = 4, GL = GR * 3 / 2;
#endif
        const unsigned long long m = T.rmask(0, k, T.anc[K]);
        const unsigned mlo = __builtin_amdgcn_readfirstlane((unsigned)m);
#pragma unroll
        for (int c = 0; c < N; c += GR) {
            const unsigned bits = (mlo >> c) & ((1u << GR) - 1u);
            if (bits) {
                double2 t[GL];
#pragma unroll
                for (int u = 0; u < GL; ++u) t[u] = *reinterpret_cast<const double2*>(&db[c * 3 + 2 * u]);
                const double* tt = reinterpret_cast<const double*>(t);
#pragma unroll
                for (int i = 0; i < GR; ++i)
                    if (c + i < N) a[c + i] += tt[3 * i] * y0 + tt[3 * i + 1] * y1 + tt[3 * i + 2] * y2;
            }
        }
    }
    MVMC_WAVE_SYNC();
    if (!on) gj = 0.0;
    db[l] = gj;
    MVMC_WAVE_SYNC();
    // |g|_inf of the Euler-space gradient, solve by solve on the whole wave (39 Euler columns: more than a half holds)
    const double gg = pairw::hsum(gj * gj);
    bool stop = false;
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        Ik1Shared& Sx = h ? S1 : S0;
        const double ge = expand(T, 0, lane, nae, Sx.tmp, Sx.xn);
        const double ginf = uni(wave_max64(fabs(ge)));
        if (lane == 0) Sx.sc[1] = ginf;
        stop = stop || ginf < gtol;
    }
    MVMC_WAVE_SYNC();
    if (l == 0) S.sc[0] = gg;
    if (stop) { *code_out = pairw::PAIR_ABORT; return; }
    double scv, tauv;
    if (!pairw::tridiag_krylov_pair<N>(a, gj, na, S.sv + SV_D, S.sv + SV_E, S.sv + SV_TAU, S.sv + SV_V0, S.tmp, S.tmp + 64, &S.sc[4], scv, tauv)) {
        *code_out = pairw::PAIR_ABORT;
        return;
    }
    double pk[(N - 2) / 2];
    pairw::pack_reflectors_pair<N>(a, pk);
    const bool dump_me = (dumpbits >> (hi ? 1 : 0)) & 1;
    if (uni(dumpbits) != 0) {
        mvmc_gdouble* hh = hh0 + (hi ? dhh : 0);
        pairw::dump_reflectors_pair<N>(pk, scv, na, hh, dump_me);
        if (dump_me) { hh[6400 + l] = refl[l]; if (l + 32 < 55) hh[6400 + 32 + l] = refl[32 + l]; }
    }
    if (!pairw::krylov_block_ok_pair(S.sv + SV_D, S.sv + SV_E, na, S.sc[6], S.tmp, S.tmp + 64)) { *code_out = pairw::PAIR_ABORT; return; }
    MVMC_WAVE_SYNC();
    // ---- the trial steps, in the tridiagonal bases ----
    const double beta0 = S.sc[4], tau0 = S.sc[5], pivmin = 1e-16 * S.sc[6] + 1e-300;
    double* rh = S.tmp;
    double* cv = S.tmp + 64;
    rh[l] = l == 0 ? beta0 : 0.0;
    MVMC_WAVE_SYNC();
    double pred, step_norm;
    const double alpha = pairw::tr_solve_tri_pair(S.sv + SV_D, S.sv + SV_E, rh, na, Delta, alpha_in, gg, pivmin, cv, &pred, &step_norm);
    MVMC_WAVE_SYNC();
    const double c = l < na ? cv[l] : 0.0;
    const double stepr = pairw::apply_q_packed_pair<N>(pk, scv, tauv, S.sv + SV_V0, tau0, na, c);
    if (l == 0) { S.sc[2] = alpha; S.sc[3] = pred; S.sc[8] = step_norm; S.sc[11] = (double)na; }
    MVMC_WAVE_SYNC();
    S.tmp[l] = on ? stepr : 0.0;
    MVMC_WAVE_SYNC();
    // back to Euler space and the trial points, solve by solve on the whole wave
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        Ik1Shared& Sx = h ? S1 : S0;
        const double stepj = expand(T, 0, lane, nae, Sx.tmp, Sx.xn);
        MVMC_WAVE_SYNC();
        ik1_trial_point(Sx, T, 0, nae, stepj);
    }
    *code_out = 1;
}

// ---------------------------------------------------------------------------------------------
// The meeting point (see the top of this file).  Called by a wave that needs a stage-1 model with evaluations left; me = 0 / 1: the
// even / odd wave of the pair, dS = bytes from this wave's solve block to the partner's, dhh = doubles from its scratch to the
// partner's.  *code_out = 1 / 2 (model and trial made, by this wave / by the partner), or PAIR_ALONE: build it the ordinary way.
// ---------------------------------------------------------------------------------------------
// (the result comes back through the caller's stack, like the model functions': see ik1_trf)
__device__ __noinline__ void ik1_pair_sync(Ik1Shared& Sin, const Ik1Tables& T, double gtol, mvmc_gdouble* __restrict__ hh, double Delta,
                                           double alpha, bool dump, int dS, int dhh, int me, int32_t* ovf, int* code_out) {
    MVMC_ASSUME_LDS(&Sin);
    MVMC_ASSUME_LDS(&T);
    Ik1Shared& S = *uni(&Sin);
    dS = uni(dS); me = uni(me); dhh = uni(dhh);
    hh = uni(hh); ovf = uni(ovf);
    const int lane = threadIdx.x & 63;
    const Ik1Pair P = {dS, dhh, me, ovf};
    int* M = ik1_pair_mailbox(S, P);
    int* C = ik1_pair_counter(S, P);
    MVMC_ASSUME_LDS(M);
    MVMC_ASSUME_LDS(C);
    // what the builder of a pair needs from this solve, whoever it turns out to be
    if (lane == 0) { S.sc[8] = Delta; S.sc[2] = alpha; }
    MVMC_WAVE_SYNC();
    const int want = 0x100 | (dump ? 2 : 0) | me;            // this wave's request (never 0, never PAIR_TAKEN)
    const int away_partner = 1 << (16 + (me ^ 1));
    auto first_lane = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    while (true) {
        int old = 0;
        const int c0 = first_lane(__hip_atomic_load(C, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) & 0xffff;
        if (lane == 0) {
            int expected = 0;
            __hip_atomic_compare_exchange_strong(M, &expected, want, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            old = expected;
        }
        old = first_lane(old);
        if (old == 0) {
            // first at the meeting: wait for the partner to take the request, or to be away
            // (every wait here ends by construction -- see the top of the file; the poll counts below are a net under that argument: a
            // wave that gives up raises bit 3 of its chain's void word, builds alone, and the host refuses the launch's results)
            bool taken = false;
            unsigned polls = 0;
            while (true) {
                const int m = first_lane(__hip_atomic_load(M, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (m != want) { taken = true; break; }
                const int c = first_lane(__hip_atomic_load(C, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (c & away_partner) {
                    int got = 0;
                    if (lane == 0) {
                        int expected = want;
                        __hip_atomic_compare_exchange_strong(M, &expected, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        got = expected;
                    }
                    got = first_lane(got);
                    if (got == want) { *code_out = pairw::PAIR_ALONE; return; }   // withdrawn
                    taken = true;                                 // taken in the meantime
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
                if (++polls > (1u << 21)) break;
            }
            if (!taken) {
                if (ovf && lane == 0) atomicOr(ovf, 8);
                *code_out = pairw::PAIR_ALONE;
                return;
            }
            // the partner builds both models: sleep until its completion count moves, then the results are in this wave's block
            polls = 0;
            while ((first_lane(__hip_atomic_load(C, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) & 0xffff) == c0) {
                __builtin_amdgcn_s_sleep(32);
                if (++polls > (1u << 20)) { if (ovf && lane == 0) atomicOr(ovf, 8); *code_out = pairw::PAIR_ALONE; return; }
            }
            MVMC_WAVE_SYNC();
            const int code = uni((int)S.sc[10]);
            *code_out = code == 1 ? 2 : pairw::PAIR_ALONE;   // (2: made by the partner)
            return;
        }
        if (old == pairw::PAIR_TAKEN || old == want) {   // (cannot happen: a pair's builder empties the mailbox before either wave comes back)
            if (ovf && lane == 0) atomicOr(ovf, 8);
            *code_out = pairw::PAIR_ALONE;
            return;
        }
        // the partner's request is waiting: take it
        int got = 0;
        if (lane == 0) {
            int expected = old;
            __hip_atomic_compare_exchange_strong(M, &expected, pairw::PAIR_TAKEN, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            got = expected;
        }
        got = first_lane(got);
        if (got != old) continue;          // withdrawn in the meantime: start over
        int code = pairw::PAIR_ABORT;
        const int dumpbits = (dump ? 1 : 0) | ((old & 2) ? 2 : 0);
        ik1_model_pair_r(S, dS, T, gtol, hh, dhh, dumpbits, &code);
        code = uni(code);
        Ik1Shared& Sp = *reinterpret_cast<Ik1Shared*>(reinterpret_cast<char*>(&S) + dS);
        MVMC_ASSUME_LDS(&Sp);
        MVMC_WAVE_SYNC();
        if (lane == 0) {
            Sp.sc[10] = (double)code;
            __hip_atomic_store(M, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(C, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        *code_out = code == 1 ? 1 : pairw::PAIR_ALONE;
        return;
    }
}
