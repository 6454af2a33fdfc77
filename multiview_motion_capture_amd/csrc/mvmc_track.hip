// Temporal layer for gfx950: match_spatial_time's node graph (AS-7/8/9) and the tracker state machine
// (TK-1) for a batch of independent chains (sub-sequences), one chain per workgroup / thread.
//   fmats_p     get_fundamental_matrix(P_i, P_j) for every view pair        mv_math_util.py:57-77
//   st_affinity distance / affinity of [tracklets | 2-D poses by view]       motion_capture.py:651-756
//               2D-2D: calc_epipolar_error (mv_math_util.py:80-115), 2D-3D: reprojection_error (:403-414)
//   assign      clusters -> IK problems (one pose per view, first wins)      motion_capture.py:763-808, :618-626
//   commit      MvTracklet.update / mark_missed / new tracklets              motion_capture.py:352-391, :924-963
#include "mvmc_common.h"

namespace {

// common joints BASIC_18 -> COCO used by reprojection_error (pose_def.py:278-288)
__device__ __constant__ const int kRpSkel[15] = {1, 2, 3, 4, 5, 6, 9, 10, 11, 12, 13, 14, 15, 16, 17};
__device__ __constant__ const int kRpCoco[15] = {11, 13, 15, 12, 14, 16, 5, 7, 9, 6, 8, 10, 0, 3, 4};

__device__ inline double det4(const double* r0, const double* r1, const double* r2, const double* r3) {
    // Laplace expansion along the first two rows
    const double s0 = r0[0] * r1[1] - r0[1] * r1[0], s1 = r0[0] * r1[2] - r0[2] * r1[0];
    const double s2 = r0[0] * r1[3] - r0[3] * r1[0], s3 = r0[1] * r1[2] - r0[2] * r1[1];
    const double s4 = r0[1] * r1[3] - r0[3] * r1[1], s5 = r0[2] * r1[3] - r0[3] * r1[2];
    const double c5 = r2[2] * r3[3] - r2[3] * r3[2], c4 = r2[1] * r3[3] - r2[3] * r3[1];
    const double c3 = r2[1] * r3[2] - r2[2] * r3[1], c2 = r2[0] * r3[3] - r2[3] * r3[0];
    const double c1 = r2[0] * r3[2] - r2[2] * r3[0], c0 = r2[0] * r3[1] - r2[1] * r3[0];
    return s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
}

// F[i][j] = det([X_j; Y_i]) with X, Y the row-pair stacks of P1, P2 (x2^T F x1 = 0)
__global__ void fmats_p_kernel(const double* __restrict__ Pm, int C, double* __restrict__ F2) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= C * C) return;
    const int a = idx / C, b = idx - a * C;
    const double* P1 = Pm + a * 12;
    const double* P2 = Pm + b * 12;
    const int rp[3][2] = {{1, 2}, {2, 0}, {0, 1}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            F2[idx * 9 + i * 3 + j] = det4(P1 + rp[j][0] * 4, P1 + rp[j][1] * 4, P2 + rp[i][0] * 4, P2 + rp[i][1] * 4);
}

// The epipolar line of a point under F (transposed = false: l = F x, the line in the other image) or under F^T, scaled as
// calc_epipolar_error scales it (by 1 / sqrt(a^2 + b^2) unless that is zero), and the norm of the SCALED (a, b) the distance divides by.
// Shared by epipolar_error and the line tables of st_pose_pairs_lines, so that both form the same expressions.
__device__ __forceinline__ void epi_line(const double* F, double x, double y, bool transposed, double& a, double& b, double& c, double& nrm) {
    if (!transposed) { a = F[0] * x + F[1] * y + F[2]; b = F[3] * x + F[4] * y + F[5]; c = F[6] * x + F[7] * y + F[8]; }
    else             { a = F[0] * x + F[3] * y + F[6]; b = F[1] * x + F[4] * y + F[7]; c = F[2] * x + F[5] * y + F[8]; }
    const double nu = a * a + b * b, sc = nu != 0.0 ? 1.0 / sqrt(nu) : 1.0;
    a *= sc; b *= sc; c *= sc;
    nrm = sqrt(a * a + b * b);
}
__device__ __forceinline__ double epi_dist(double a, double b, double c, double nrm, double x, double y) { return fabs(a * x + b * y + c) / nrm; }

// mean over valid joints of the symmetric point-to-epiline distance; NaN if none is valid (epi_line / epi_dist above are these
// expressions, factored out for the line tables; tests/test_gpu_config5_c8p8.py compares the two forms bit for bit)
__device__ __noinline__ double epipolar_error(const double* F, const double* k1, const double* k2, double min_score) {
    double total = 0.0;
    int cnt = 0;
    for (int j = 0; j < 17; ++j) {
        const double x1 = k1[j * 3], y1 = k1[j * 3 + 1], x2 = k2[j * 3], y2 = k2[j * 3 + 1];
        if (!(k1[j * 3 + 2] * k2[j * 3 + 2] > min_score)) continue;
        // image 1 -> line in image 2: l = F x1
        double a = F[0] * x1 + F[1] * y1 + F[2], b = F[3] * x1 + F[4] * y1 + F[5], c = F[6] * x1 + F[7] * y1 + F[8];
        double nu = a * a + b * b, sc = nu != 0.0 ? 1.0 / sqrt(nu) : 1.0;
        a *= sc; b *= sc; c *= sc;
        const double d1 = fabs(a * x2 + b * y2 + c) / sqrt(a * a + b * b);
        // image 2 -> line in image 1: l = F^T x2
        a = F[0] * x2 + F[3] * y2 + F[6]; b = F[1] * x2 + F[4] * y2 + F[7]; c = F[2] * x2 + F[5] * y2 + F[8];
        nu = a * a + b * b; sc = nu != 0.0 ? 1.0 / sqrt(nu) : 1.0;
        a *= sc; b *= sc; c *= sc;
        const double d2 = fabs(a * x1 + b * y1 + c) / sqrt(a * a + b * b);
        total = total + 0.5 * (d1 + d2);
        ++cnt;
    }
    return cnt ? total / cnt : __longlong_as_double(0x7ff8000000000000LL);
}

__device__ __noinline__ double reproj_error(const double* joints, const double* k2, const double* P, double min_score) {
    double total = 0.0;
    int cnt = 0;
    for (int m = 0; m < 15; ++m) {
        const double* X = joints + kRpSkel[m] * 3;
        const double* kp = k2 + kRpCoco[m] * 3;
        if (!(kp[2] > min_score)) continue;  // 3-D scores are ones
        const double h0 = P[0] * X[0] + P[1] * X[1] + P[2] * X[2] + P[3];
        const double h1 = P[4] * X[0] + P[5] * X[1] + P[6] * X[2] + P[7];
        const double h2 = P[8] * X[0] + P[9] * X[1] + P[10] * X[2] + P[11];
        const double du = h0 / (1e-5 + h2) - kp[0], dv = h1 / (1e-5 + h2) - kp[1];
        total += sqrt(du * du + dv * dv);
        ++cnt;
    }
    return cnt ? total / cnt : __longlong_as_double(0x7ff8000000000000LL);
}

// One chain-frame (chain b, frame f) on the calling wave (NT = 64) or on the whole NT-thread workgroup (NT = 256 / 512:
// every thread must call).  sm: (NS*NS + 10) doubles + 2*NS ints of LDS, NS = T + C*P.
// The 2-D / 2-D block of the match_spatial_time graph (calc_epipolar_error per pose pair of different views, motion_capture.py:686-700)
// depends only on the frame's own detections: a chain-frame workgroup computes it while it waits for its predecessor's tracklets.
// E[q_i * N + q_j], q = view * P + person, N = C * P; entries of absent poses and same-view pairs are not written (never read).
// The frame's keypoints (C * P poses x 51 doubles) into LDS, one batch of coalesced loads; every thread of the workgroup calls, the
// caller synchronises.  epipolar_error / reproj_error walk a pose's 17 joints with a data-dependent `continue` per joint: from global
// memory that is a chain of dependent memory round trips per pose pair (the pattern that made up the triangulation kernel's 2.7 ms,
// DESIGN.md section 6); from LDS it is arithmetic.
__device__ __forceinline__ void st_stage_keypoints(double* dst, const double* __restrict__ kps17, int f, int C, int P) {
    const double* src = kps17 + (size_t)f * C * P * 51;
    for (int e = threadIdx.x; e < C * P * 51; e += blockDim.x) dst[e] = src[e];
}
// kf_lds: the frame's keypoints staged in LDS by the caller (st_stage_keypoints), or nullptr = read them from kps17
__device__ __forceinline__ void st_pose_pairs(double* E, const double* __restrict__ kps17, const int32_t* __restrict__ counts, int f,
                                              const double* __restrict__ F2, int C, int P, double min_score, const double* kf_lds = nullptr) {
    const int N = C * P;
    const double* kf = kf_lds ? kf_lds : kps17 + (size_t)f * C * P * 51;
    for (int e = threadIdx.x; e < N * N; e += blockDim.x) {
        const int qi = e / N, qj = e - qi * N;
        const int ci = qi / P, cj = qj / P;
        if (ci == cj) continue;
        int ni = counts[f * C + ci], nj = counts[f * C + cj];
        ni = ni < 0 ? 0 : (ni > P ? P : ni);
        nj = nj < 0 ? 0 : (nj > P ? P : nj);
        if (qi - ci * P >= ni || qj - cj * P >= nj) continue;
        E[e] = epipolar_error(F2 + (ci * C + cj) * 9, kf + qi * 51, kf + qj * 51, min_score);
    }
}

// The same block E by LINE TABLES (the BIG layout: 64 poses, 3,584 ordered pairs per frame, where the pair errors were 11 % of a
// chain's cycles).  A pair's error needs, per joint, the epipolar line of pose i's joint in the other image and that of pose j's joint
// -- a square root and a division to scale each, another square root for the distance's divisor -- and a line depends on ONE pose and
// the pair of VIEWS only: the P poses of the other view share it.  Per unordered pair of views {u, v} (two of them per round: the tables
// fit st_affinity_wave's part of the scratch, which is idle until the block is complete): four tables -- poses of u under F_uv, poses
// of v under F_uv^T, poses of v under F_vu, poses of u under F_vu^T (the two fundamental matrices are made independently, so (i, j) and
// (j, i) are different numbers, as in the reference, motion_capture.py:672-700) -- then one thread per ordered pair sums the joints in
// order with one division per line where epipolar_error does two square roots and two divisions: the same expressions on the same
// operands (epi_line / epi_dist), bit for bit, in a third of the square roots and divisions.
// Ls: 2 * 4 * P * 68 doubles of LDS; kf: the frame's keypoints in LDS; every thread of the workgroup calls; ends synchronised.
template <int NT>
__device__ __forceinline__ void st_pose_pairs_lines(double* E, double* Ls, const double* kf, const int32_t* __restrict__ counts, int f,
                                                    const double* __restrict__ F2, int C, int P, double min_score) {
    const int tid = threadIdx.x, N = C * P, nvp = C * (C - 1) / 2;
    auto unrank = [&](int vp, int& u, int& v) {        // vp-th pair u < v in lexicographic order
        u = 0;
        for (int n = C - 1; vp >= n; --n) { vp -= n; ++u; }
        v = u + 1 + vp;
    };
    for (int r0 = 0; r0 < nvp; r0 += 2) {
        for (int it = tid; it < 2 * 4 * P * 17; it += NT) {
            const int joint = it % 17;
            int t = it / 17;
            const int p = t % P;
            t /= P;
            const int set = t & 3, vl = t >> 2, vp = r0 + vl;
            if (vp >= nvp) continue;
            int u, v;
            unrank(vp, u, v);
            const int view = (set == 0 || set == 3) ? u : v;
            const double* Fm = F2 + (set < 2 ? u * C + v : v * C + u) * 9;
            const double* k = kf + (view * P + p) * 51 + joint * 3;
            double a, b, c, nrm;
            epi_line(Fm, k[0], k[1], (set & 1) != 0, a, b, c, nrm);
            double* L = Ls + (((vl * 4 + set) * P + p) * 17 + joint) * 4;
            L[0] = a; L[1] = b; L[2] = c; L[3] = nrm;
        }
        __syncthreads();
        for (int it = tid; it < 2 * 2 * P * P; it += NT) {
            const int pj = it % P;
            int t = it / P;
            const int pi = t % P;
            t /= P;
            const int dir = t & 1, vl = t >> 1, vp = r0 + vl;
            if (vp >= nvp) continue;
            int u, v;
            unrank(vp, u, v);
            const int va = dir ? v : u, vb = dir ? u : v;      // the ordered pair: first pose in view va, second in vb; F = F2[va][vb]
            int na = counts[f * C + va], nb = counts[f * C + vb];
            na = na < 0 ? 0 : (na > P ? P : na);
            nb = nb < 0 ? 0 : (nb > P ? P : nb);
            if (pi >= na || pj >= nb) continue;
            const int qi = va * P + pi, qj = vb * P + pj;
            const double* L1 = Ls + ((vl * 4 + (dir ? 2 : 0)) * P + pi) * 68;    // lines of the first pose's joints under F
            const double* L2 = Ls + ((vl * 4 + (dir ? 3 : 1)) * P + pj) * 68;    // lines of the second pose's joints under F^T
            const double* k1 = kf + qi * 51, * k2 = kf + qj * 51;
            double total = 0.0;
            int cnt = 0;
            for (int j = 0; j < 17; ++j) {
                if (!(k1[j * 3 + 2] * k2[j * 3 + 2] > min_score)) continue;
                const double d1 = epi_dist(L1[j * 4], L1[j * 4 + 1], L1[j * 4 + 2], L1[j * 4 + 3], k2[j * 3], k2[j * 3 + 1]);
                const double d2 = epi_dist(L2[j * 4], L2[j * 4 + 1], L2[j * 4 + 2], L2[j * 4 + 3], k1[j * 3], k1[j * 3 + 1]);
                total = total + 0.5 * (d1 + d2);
                ++cnt;
            }
            E[qi * N + qj] = cnt ? total / cnt : __longlong_as_double(0x7ff8000000000000LL);
        }
        __syncthreads();
    }
}

template <int NT>
__device__ __forceinline__ void st_affinity_wave(double* sm, const double* __restrict__ kps17, const int32_t* __restrict__ counts,
                                                 int b, int f, const double* __restrict__ track_joints,
                                                 const int32_t* __restrict__ n_tracks, const double* __restrict__ Pm,
                                                 const double* __restrict__ F2, int C, int P, int T, double min_score,
                                                 double* __restrict__ W, double* __restrict__ Dout,
                                                 int32_t* __restrict__ group_counts, const double* Epre = nullptr, int ldE = 0,
                                                 const double* kf_lds = nullptr) {
    // Epre: the pose-pair epipolar errors of this frame made ahead of time (st_pose_pairs; [q_i * ldE + q_j], q = view * P + person)
    constexpr bool WG = NT > 64;
    const int tid = WG ? (int)threadIdx.x : (int)(threadIdx.x & 63);
    auto sync = [] { if constexpr (WG) __syncthreads(); else MVMC_WAVE_SYNC(); };
    const int NS = T + C * P;
    double* D = sm;                       // [NS*NS]
    double& s_max = D[NS * NS];
    double* s_part = D + NS * NS + 2;     // per-wave maxima (up to 8 waves)
    int* nview = reinterpret_cast<int*>(D + NS * NS + 10);  // node -> view (-1 = tracklet)
    int* nidx = nview + NS;               // node -> tracklet slot or local pose index c*P+p
    int& s_n = reinterpret_cast<int*>(D + NS * NS + 1)[0];
    int nt = mvmc_ld_i32(n_tracks + b);
    nt = nt < 0 ? 0 : (nt > T ? T : nt);
    if (tid == 0) {
        int n = 0;
        for (int t = 0; t < nt; ++t) { nview[n] = -1; nidx[n] = t; ++n; }
        group_counts[b * (C + 1)] = nt;
        for (int c = 0; c < C; ++c) {
            int cnt = counts[f * C + c];
            cnt = cnt < 0 ? 0 : (cnt > P ? P : cnt);
            if (nt == 0) cnt = 0;  // chains without tracklets take the match_spatial path
            group_counts[b * (C + 1) + 1 + c] = cnt;
            for (int p = 0; p < cnt; ++p) { nview[n] = c; nidx[n] = c * P + p; ++n; }
        }
        s_n = n;
    }
    sync();
    const int n = s_n;
    const double* kf = kf_lds ? kf_lds : kps17 + (size_t)f * C * P * 51;
    const double* tj = track_joints + (size_t)b * T * 54;
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    for (int e = tid; e < n * n; e += NT) {
        const int i = e / n, j = e - i * n;
        double d;
        if (i == j) d = 0.0;
        else {
            const int vi = nview[i], vj = nview[j];
            if (vi >= 0 && vi == vj) d = nan;
            else if (vi >= 0 && vj >= 0)
                d = Epre ? Epre[nidx[i] * ldE + nidx[j]]
                         : epipolar_error(F2 + (vi * C + vj) * 9, kf + nidx[i] * 51, kf + nidx[j] * 51, min_score);
            else if (vi >= 0) d = reproj_error(tj + nidx[j] * 54, kf + nidx[i] * 51, Pm + vi * 12, min_score);
            else if (vj >= 0) d = reproj_error(tj + nidx[i] * 54, kf + nidx[j] * 51, Pm + vj * 12, min_score);
            else d = nan;
        }
        D[e] = d;
    }
    sync();
    // nanmax, NaN -> max + 1 (motion_capture.py:744-745)
    double m = -1e300;
    for (int e = tid; e < n * n; e += NT) { const double d = D[e]; if (d == d && d > m) m = d; }
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(m, off, 64); m = o > m ? o : m; }
    if ((tid & 63) == 0) s_part[tid >> 6] = m;
    sync();
    if (tid == 0) {
        double mm = s_part[0];
        if constexpr (WG) { for (int w = 1; w < NT / 64; ++w) mm = s_part[w] > mm ? s_part[w] : mm; }
        s_max = mm;
    }
    sync();
    double* Wb = W + (size_t)b * NS * NS;
    double* Db = Dout ? Dout + (size_t)b * NS * NS : nullptr;
    for (int e = tid; e < NS * NS; e += NT) {
        const int i = e / NS, j = e - i * NS;
        double d = 0.0, s = 0.0, raw = 0.0;
        if (i < n && j < n) {
            d = raw = D[i * n + j];
            if (!(d == d)) d = s_max + 1.0;
            s = 1.0 / (1.0 + exp(5.0 * ((d - 15.0) / 30.0)));
            if (s < 1e-3) s = 0.0;
            if (s > 1.0) s = 1.0;
        }
        Wb[e] = s;
        if (Db) Db[e] = raw;  // raw distances (NaN where the reference's matrix holds NaN before its nanmax fill)
    }
}

__global__ void __launch_bounds__(64)
st_affinity_kernel(const double* __restrict__ kps17, const int32_t* __restrict__ counts,
                   const int32_t* __restrict__ frame_idx, const double* __restrict__ track_joints,
                   const int32_t* __restrict__ n_tracks, const double* __restrict__ Pm, const double* __restrict__ F2,
                   int C, int P, int T, double min_score, double* __restrict__ W, double* __restrict__ Dout,
                   int32_t* __restrict__ group_counts) {
    extern __shared__ double sm[];
    const int b = blockIdx.x;
    st_affinity_wave<64>(sm, kps17, counts, b, frame_idx[b], track_joints, n_tracks, Pm, F2, C, P, T, min_score, W, Dout,
                     group_counts);
}

// One thread per chain: cluster labels -> IK problem descriptors.
//   slots [0,T): live tracklets (warm, init = their parameters) -- status 0 unmatched, 1 one view, 2 update
//   slots [T,T+K): new tracklets from 2-D-only clusters with >= 2 views (cold)
__device__ __forceinline__ void assign_chain(int lane, int nl, int b, int f, const int32_t* __restrict__ labels_sp,
                                             const int32_t* __restrict__ ncl_sp, const int32_t* __restrict__ labels_st,
                                             const int32_t* __restrict__ ncl_st, const int32_t* __restrict__ counts,
                                             const int32_t* __restrict__ n_tracks, const double* __restrict__ track_params,
                                             int C, int P, int T, int K, int V, int32_t* __restrict__ members,
                                             uint8_t* __restrict__ cold, double* __restrict__ init,
                                             int32_t* __restrict__ status, int32_t* __restrict__ n_new, int32_t* ovf = nullptr,
                                             int32_t* __restrict__ n_members = nullptr) {
    // ovf (one word, may be shared by several chains: atomic OR): bit 0 = a cluster or a member was dropped here for lack of room
    // (more than K new tracklets, more than V views) -- the reference has no such caps, so the results of the frame are not its own.
    // With K >= (poses of a frame) / 2 and V >= the poses of a frame neither can happen: a new tracklet needs two poses, and the
    // clusters are disjoint (what the callers pass by default).
    // n_members (B, T + K) or NULL: with it, the member count of every problem slot is written there and the member table is NOT
    // padded with -1 (the chain kernel reads the counts; the table's rows are then valid in [0, count) only)
    int nt = mvmc_ld_i32(n_tracks + b);
    nt = nt < 0 ? 0 : (nt > T ? T : nt);
    const int NP = T + K;
    int32_t* mem = members + (size_t)b * NP * V;
    // bulk initialisation, spread over the nl lanes that call (lane = 0 .. nl-1; the cluster logic below is lane 0's)
    int32_t* nmem = n_members ? n_members + (size_t)b * NP : nullptr;
    if (nmem) { for (int s = lane; s < NP; s += nl) nmem[s] = 0; }
    else for (int e = lane; e < NP * V; e += nl) mem[e] = -1;
    for (int s = lane; s < NP; s += nl) cold[(size_t)b * NP + s] = s >= T;
    for (int s = lane; s < T; s += nl) status[(size_t)b * T + s] = 0;
    for (int e = lane; e < T * 68; e += nl) {
        const int s = e / 68;
        init[(size_t)b * NP * 68 + e] = s < nt ? track_params[(size_t)b * T * 68 + e] : 0.0;
    }
    if (nl > 1) MVMC_WAVE_SYNC();   // (callers with several lanes pass lanes of ONE wave)
    if (nl == 64) {
        // ---- a full wave: the cluster logic on ballot masks (lane = graph node; nodes 64 .. 127 in a second word) ----
        // The same decisions as lane 0's walk below (which the one-thread kernel still takes), but every test over the nodes is one
        // v_cmp + scalar bit operations instead of a loop of dependent scalar-style code on one lane: that walk was 1.5 % of a chain's
        // cycles (~58 k cycles per frame for ~120 inner iterations).
        int cntw = 0;
        if (lane < C) { const int k = counts[f * C + lane]; cntw = k < 0 ? 0 : (k > P ? P : k); }
        int start[17];      // (static indices only: the loops over the views are unrolled, so this stays in scalar registers)
        start[0] = 0;
#pragma unroll
        for (int c = 0; c < 16; ++c) start[c + 1] = start[c] + (c < C ? __builtin_amdgcn_readlane(cntw, c) : 0);
        const int n_pose = start[16], n_nodes = nt + n_pose;
        const int32_t* labp = (nt == 0) ? labels_sp + (size_t)b * C * P : labels_st + (size_t)b * (T + C * P);
        // per node (two per lane): label, view, index in the view, the value a member entry gets, the mask of earlier nodes of its view
        int lab2[2], val2[2];
        unsigned long long same_lo[2], same_hi[2];
        bool pose2[2];
        for (int h = 0; h < 2; ++h) {
            const int node = lane + 64 * h, idx = node - nt;
            lab2[h] = node < n_nodes ? labp[node] : -1;
            pose2[h] = idx >= 0 && idx < n_pose;
            int cv = 0, base = 0;
#pragma unroll
            for (int c = 1; c < 16; ++c)
                if (c < C && idx >= start[c]) { cv = c; base = start[c]; }
            const int pv = idx - base;
            val2[h] = (f * C + cv) * P + pv;
            // earlier nodes of the same view: [nt + base, node)
            const int r0 = nt + base, r1 = node;
            auto below = [](int bit) -> unsigned long long { return bit <= 0 ? 0ull : (bit >= 64 ? ~0ull : ((1ull << bit) - 1ull)); };
            same_lo[h] = below(r1) & ~below(r0);
            same_hi[h] = below(r1 - 64) & ~below(r0 - 64);
        }
        const unsigned long long lower_lo = (lane == 0) ? 0ull : ((1ull << lane) - 1ull);   // nodes below this lane's (per word)
        const int nc = (nt == 0) ? ncl_sp[b] : ncl_st[b];
        int created = 0;
        for (int k = 0; k < nc; ++k) {
            const unsigned long long in_lo = __builtin_amdgcn_ballot_w64(lab2[0] == k), in_hi = __builtin_amdgcn_ballot_w64(lab2[1] == k);
            // members: every pose node of the cluster (match_spatial), the first pose node of each view (match_spatial_time)
            bool mem2[2];
            for (int h = 0; h < 2; ++h) {
                mem2[h] = lab2[h] == k && pose2[h];
                if (nt != 0) mem2[h] = mem2[h] && ((in_lo & same_lo[h]) | (in_hi & same_hi[h])) == 0ull;
            }
            const unsigned long long m_lo = __builtin_amdgcn_ballot_w64(mem2[0]), m_hi = __builtin_amdgcn_ballot_w64(mem2[1]);
            int m = __popcll(m_lo) + __popcll(m_hi);
            if (nt == 0 && m > 64) { if (ovf && lane == 0) atomicOr(ovf, 1); m = 64; }   // (a cluster of more than 64 poses: cut, and SAID so)
            const int rank0 = __popcll(m_lo & lower_lo), rank1 = __popcll(m_lo) + __popcll(m_hi & lower_lo);
            int tracklet = -1;
            if (nt != 0) {
                const unsigned long long tm = in_lo & ((1ull << nt) - 1ull);
                if (tm) tracklet = __builtin_ctzll(tm);
            }
            int slot = -1;       // row of the member table this cluster is written to
            if (nt == 0) {
                if (m >= 2) {
                    if (ovf && m > V && lane == 0) atomicOr(ovf, 1);
                    if (created < K) { slot = T + created; ++created; }
                    else if (ovf && lane == 0) atomicOr(ovf, 1);
                }
            } else {
                if (ovf && m >= 2 && m > V && lane == 0) atomicOr(ovf, 1);
                if (tracklet >= 0) {
                    if (m > 0) {
                        if (lane == 0) status[(size_t)b * T + tracklet] = m >= 2 ? 2 : 1;
                        if (m >= 2) slot = tracklet;
                    }
                } else if (m >= 2) {
                    if (created < K) { slot = T + created; ++created; }
                    else if (ovf && lane == 0) atomicOr(ovf, 1);
                }
            }
            if (slot >= 0) {
                if (mem2[0] && rank0 < V) mem[slot * V + rank0] = val2[0];
                if (mem2[1] && rank1 < V) mem[slot * V + rank1] = val2[1];
                if (nmem && lane == 0) nmem[slot] = m < V ? m : V;
            }
        }
        if (lane == 0) n_new[b] = created;
        return;
    }
    if (lane != 0) return;
    int cnt[16];
    for (int c = 0; c < C && c < 16; ++c) {
        int k = counts[f * C + c];
        cnt[c] = k < 0 ? 0 : (k > P ? P : k);
    }
    int created = 0;
    if (nt == 0) {
        // match_spatial path: node order = 2-D poses by view; every member is kept (motion_capture.py:621-626)
        const int32_t* lab = labels_sp + (size_t)b * C * P;
        const int nc = ncl_sp[b];
        for (int k = 0; k < nc; ++k) {
            int m = 0, node = 0;
            int32_t tmp[64];
            for (int c = 0; c < C; ++c)
                for (int p = 0; p < cnt[c]; ++p, ++node)
                    if (lab[node] == k && m < 64) tmp[m++] = (f * C + c) * P + p;
            if (m >= 2) {
                if (ovf && m > V) atomicOr(ovf, 1);
                if (created < K) {
                    for (int v = 0; v < m && v < V; ++v) mem[(T + created) * V + v] = tmp[v];
                    if (nmem) nmem[T + created] = m < V ? m : V;
                    ++created;
                } else if (ovf) atomicOr(ovf, 1);
            }
        }
    } else {
        const int NS = T + C * P;
        const int32_t* lab = labels_st + (size_t)b * NS;
        const int nc = ncl_st[b];
        for (int k = 0; k < nc; ++k) {
            int tracklet = -1;
            for (int t = 0; t < nt; ++t)
                if (lab[t] == k) { tracklet = t; break; }
            int m = 0, node = nt;
            int32_t tmp[16];
            for (int c = 0; c < C; ++c) {
                bool used = false;
                for (int p = 0; p < cnt[c]; ++p, ++node)
                    if (lab[node] == k && !used && m < 16) { tmp[m++] = (f * C + c) * P + p; used = true; }
            }
            if (ovf && m >= 2 && m > V) atomicOr(ovf, 1);
            if (tracklet >= 0) {
                if (m > 0) {
                    status[(size_t)b * T + tracklet] = m >= 2 ? 2 : 1;
                    if (m >= 2) {
                        for (int v = 0; v < m && v < V; ++v) mem[tracklet * V + v] = tmp[v];
                        if (nmem) nmem[tracklet] = m < V ? m : V;
                    }
                }
            } else if (m >= 2) {
                if (created < K) {
                    for (int v = 0; v < m && v < V; ++v) mem[(T + created) * V + v] = tmp[v];
                    if (nmem) nmem[T + created] = m < V ? m : V;
                    ++created;
                } else if (ovf) atomicOr(ovf, 1);
            }
        }
    }
    n_new[b] = created;
}

__global__ void assign_kernel(const int32_t* __restrict__ labels_sp, const int32_t* __restrict__ ncl_sp,
                              const int32_t* __restrict__ labels_st, const int32_t* __restrict__ ncl_st,
                              const int32_t* __restrict__ counts, const int32_t* __restrict__ frame_idx,
                              const int32_t* __restrict__ n_tracks, const double* __restrict__ track_params, int B,
                              int C, int P, int T, int K, int V, int32_t* __restrict__ members,
                              uint8_t* __restrict__ cold, double* __restrict__ init, int32_t* __restrict__ status,
                              int32_t* __restrict__ n_new, int32_t* __restrict__ overflow) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    assign_chain(0, 1, b, frame_idx[b], labels_sp, ncl_sp, labels_st, ncl_st, counts, n_tracks, track_params, C, P, T, K, V, members,
                 cold, init, status, n_new, overflow ? overflow + b : nullptr);
}

// One thread per chain: tracklet table after the frame's IK solves.
// meta[b][slot] = {id, state (1 tentative, 2 confirmed), hits, length}
__device__ __forceinline__ void commit_chain(int lane, int nl, int b, const int32_t* __restrict__ status, const int32_t* __restrict__ n_new,
                                             const double* __restrict__ ik_params, const double* __restrict__ ik_joints, int T,
                                             int K, int n_inits, double* __restrict__ track_params,
                                             double* __restrict__ track_joints, int32_t* __restrict__ meta,
                                             int32_t* __restrict__ n_tracks, int32_t* __restrict__ next_id,
                                             int32_t* __restrict__ n_dead, int32_t* __restrict__ slot_src, int32_t* ovf = nullptr) {
    // ovf bit 1: a new tracklet did not fit the table of T slots
    const int NP = T + K;
    int nt = mvmc_ld_i32(n_tracks + b);
    nt = nt < 0 ? 0 : (nt > T ? T : nt);
    double* tp = track_params + (size_t)b * T * 68;
    double* tj = track_joints + (size_t)b * T * 54;
    int32_t* mt = meta + (size_t)b * T * 4;
    int w = 0, dead = 0;
    for (int s = 0; s < nt; ++s) {
        const int st = status[(size_t)b * T + s];
        if (st == 0) { ++dead; continue; }  // mark_missed: max_age = 0, any miss kills
        int id = mt[s * 4], state = mt[s * 4 + 1], hits = mt[s * 4 + 2], len = mt[s * 4 + 3];
        const double* sp = tp + s * 68;
        const double* sj = tj + s * 54;
        if (st == 2) {
            sp = ik_params + ((size_t)b * NP + s) * 68;
            sj = ik_joints + ((size_t)b * NP + s) * 54;
            ++hits; ++len;
            if (state == 1 && hits >= n_inits) state = 2;
        }
        // compaction only moves entries to lower slots (w <= s), and slot s's IK result is read from the
        // separate IK buffers, so the in-place copy is safe
        // every calling lane runs the same decisions; the row copies are spread over the lanes (rows move to lower or
        // equal slots only, and a later source row is never an earlier destination)
        if (w != s || st == 2) {
            for (int e = lane; e < 68; e += nl) tp[w * 68 + e] = sp[e];
            for (int e = lane; e < 54; e += nl) tj[w * 54 + e] = sj[e];
        }
        if (nl > 1) MVMC_WAVE_SYNC();
        if (lane == 0) {
            mt[w * 4] = id; mt[w * 4 + 1] = state; mt[w * 4 + 2] = hits; mt[w * 4 + 3] = len;
            if (slot_src) slot_src[(size_t)b * T + w] = (st == 2) ? s : -1;
        }  // IK problem slot solved this frame
        ++w;
    }
    int id = mvmc_ld_i32(next_id + b);
    const int nn = n_new[b];
    for (int k = 0; k < nn; ++k) {
        if (w >= T) { if (ovf && lane == 0) atomicOr(ovf, 2); break; }  // table full: dropped, and reported
        const double* sp = ik_params + ((size_t)b * NP + T + k) * 68;
        const double* sj = ik_joints + ((size_t)b * NP + T + k) * 54;
        for (int e = lane; e < 68; e += nl) tp[w * 68 + e] = sp[e];
        for (int e = lane; e < 54; e += nl) tj[w * 54 + e] = sj[e];
        if (lane == 0) {
            mt[w * 4] = id; mt[w * 4 + 1] = 1; mt[w * 4 + 2] = 1; mt[w * 4 + 3] = 1;
            if (slot_src) slot_src[(size_t)b * T + w] = T + k;
        }
        ++id;
        ++w;
    }
    if (lane != 0) return;
    if (slot_src)
        for (int s = w; s < T; ++s) slot_src[(size_t)b * T + s] = -1;
    next_id[b] = id;
    n_tracks[b] = w;
    n_dead[b] = mvmc_ld_i32(n_dead + b) + dead;
}

__global__ void commit_kernel(const int32_t* __restrict__ status, const int32_t* __restrict__ n_new,
                              const double* __restrict__ ik_params, const double* __restrict__ ik_joints, int B, int T,
                              int K, int n_inits, double* __restrict__ track_params, double* __restrict__ track_joints,
                              int32_t* __restrict__ meta, int32_t* __restrict__ n_tracks, int32_t* __restrict__ next_id,
                              int32_t* __restrict__ n_dead, int32_t* __restrict__ slot_src, int32_t* __restrict__ overflow) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    commit_chain(0, 1, b, status, n_new, ik_params, ik_joints, T, K, n_inits, track_params, track_joints, meta, n_tracks, next_id,
                 n_dead, slot_src, overflow ? overflow + b : nullptr);
}

}  // namespace

#ifndef MVMC_DEVICE_ONLY   // (mvmc_chain.hip includes the device code above)
extern "C" int mvmc_fmats_from_projections(const double* Pmats, int n_views, double* F2, mvmcStream_t stream) {
    if (!Pmats || !F2 || n_views <= 0) return MVMC_ERR_ARG;
    const int n = n_views * n_views;
    hipLaunchKernelGGL(fmats_p_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, Pmats, n_views, F2);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}

extern "C" int mvmc_st_affinity(const double* kps17, const int32_t* counts, const int32_t* frame_idx,
                                const double* track_joints, const int32_t* n_tracks, const double* Pmats,
                                const double* F2, int n_chains, int n_views, int p_max, int t_max, double min_score,
                                double* W, double* D, int32_t* group_counts, mvmcStream_t stream) {
    if (!kps17 || !counts || !frame_idx || !track_joints || !n_tracks || !Pmats || !F2 || !W || !group_counts)
        return MVMC_ERR_ARG;
    if (n_views <= 0 || p_max <= 0 || t_max <= 0) return MVMC_ERR_ARG;
    const int NS = t_max + n_views * p_max;
    if (NS > MVMC_MAX_NODES) return MVMC_ERR_UNSUPPORTED;
    if (n_chains <= 0) return n_chains == 0 ? MVMC_OK : MVMC_ERR_ARG;
    const size_t shm = (size_t)(NS * NS + 10) * sizeof(double) + (size_t)2 * NS * sizeof(int);
    hipLaunchKernelGGL(st_affinity_kernel, dim3(n_chains), dim3(64), shm, (hipStream_t)stream, kps17, counts, frame_idx,
                       track_joints, n_tracks, Pmats, F2, n_views, p_max, t_max, min_score, W, D, group_counts);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}

extern "C" int mvmc_track_assign(const int32_t* labels_sp, const int32_t* ncl_sp, const int32_t* labels_st,
                                 const int32_t* ncl_st, const int32_t* counts, const int32_t* frame_idx,
                                 const int32_t* n_tracks, const double* track_params, int n_chains, int n_views,
                                 int p_max, int t_max, int k_max, int v_max, int32_t* members, uint8_t* cold,
                                 double* init_params, int32_t* status, int32_t* n_new, int32_t* overflow, mvmcStream_t stream) {
    if (!labels_sp || !ncl_sp || !labels_st || !ncl_st || !counts || !frame_idx || !n_tracks || !track_params ||
        !members || !cold || !init_params || !status || !n_new)
        return MVMC_ERR_ARG;
    if (n_views <= 0 || n_views > 16 || p_max <= 0 || t_max <= 0 || k_max <= 0 || v_max <= 0) return MVMC_ERR_ARG;
    if (n_views * p_max > MVMC_MAX_NODES) return MVMC_ERR_UNSUPPORTED;
    if (n_chains <= 0) return n_chains == 0 ? MVMC_OK : MVMC_ERR_ARG;
    hipLaunchKernelGGL(assign_kernel, dim3((n_chains + 63) / 64), dim3(64), 0, (hipStream_t)stream, labels_sp, ncl_sp,
                       labels_st, ncl_st, counts, frame_idx, n_tracks, track_params, n_chains, n_views, p_max, t_max,
                       k_max, v_max, members, cold, init_params, status, n_new, overflow);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}

extern "C" int mvmc_track_commit(const int32_t* status, const int32_t* n_new, const double* ik_params,
                                 const double* ik_joints, int n_chains, int t_max, int k_max, int n_inits,
                                 double* track_params, double* track_joints, int32_t* meta, int32_t* n_tracks,
                                 int32_t* next_id, int32_t* n_dead, int32_t* slot_src, int32_t* overflow, mvmcStream_t stream) {
    if (!status || !n_new || !ik_params || !ik_joints || !track_params || !track_joints || !meta || !n_tracks ||
        !next_id || !n_dead)
        return MVMC_ERR_ARG;
    if (t_max <= 0 || k_max <= 0) return MVMC_ERR_ARG;
    if (n_chains <= 0) return n_chains == 0 ? MVMC_OK : MVMC_ERR_ARG;
    hipLaunchKernelGGL(commit_kernel, dim3((n_chains + 63) / 64), dim3(64), 0, (hipStream_t)stream, status, n_new,
                       ik_params, ik_joints, n_chains, t_max, k_max, n_inits, track_params, track_joints, meta,
                       n_tracks, next_id, n_dead, slot_src, overflow);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}
#endif  // MVMC_DEVICE_ONLY
