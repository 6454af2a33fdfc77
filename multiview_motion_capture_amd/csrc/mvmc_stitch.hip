// Multi-GPU glue of the frame-sharded path.  The reference has no counterpart: its tracker is one
// sequential pass over the sequence (motion_capture.py:1062-1116).  Here a sequence is cut into chains (sub-sequences that cold-start,
// DESIGN.md section 7), contiguous chain ranges go to the GPUs, and the identities are stitched across ALL chain boundaries in two
// levels, so that a rank's work does not grow with the number of ranks:
//
//   pack_tracks   (before the gather, on the shard's own data) a shard's per-frame tracklet tables -> one message: only live
//                 tracklets, float32, plus the first / last frame table of every chain; THEN the shard's own chain boundaries are
//                 matched (a wave per boundary: optimal assignment on the mean joint distance, pairs farther than max_dist dropped)
//                 and its identities numbered locally (pointer jumping along the matches, roots in chain order) -- both go into the
//                 message's `local` section
//   stitch        (after the ONE all-gather) only the world - 1 shard boundaries are matched; a serial pass over the shards (one
//                 wave, <= world x id_cap steps) turns local into global numbers; one thread per (chain, local id) writes the table
//
// The result is the one a single pass over all chain boundaries gives (oracle/stitch_np.py is written that way): roots numbered in
// chain order, identities inherited along the matches.
#include "mvmc_common.h"

namespace {

constexpr int ST_T = 16;        // max tracklet slots per frame (t_max <= 16)
constexpr int ST_HDR = 8;       // header words
constexpr double MVMC_STITCH_NO_MATCH = 1e30;

constexpr int ST_LOCAL_HDR = 8; // words in front of the local section: {local roots, matched pairs, error word, void word, 0...}

struct MsgLayout {
    int b_cap, t_max, row_cap, id_cap;
    __host__ __device__ size_t off_ids() const { return ST_HDR; }
    __host__ __device__ size_t off_bounds() const { return off_ids() + (size_t)b_cap; }
    __host__ __device__ size_t off_rows() const { return off_bounds() + (size_t)b_cap * 2 * t_max * MVMC_BOUND_WORDS; }
    // the shard's own stitch: header, lmatch (b_cap, t_max), lgid (b_cap, id_cap)
    __host__ __device__ size_t off_local() const { return off_rows() + (size_t)row_cap * MVMC_ROW_WORDS; }
    __host__ __device__ size_t off_lmatch() const { return off_local() + ST_LOCAL_HDR; }
    __host__ __device__ size_t off_lgid() const { return off_lmatch() + (size_t)b_cap * t_max; }
    __host__ __device__ size_t words() const { return off_lgid() + (size_t)b_cap * id_cap; }
};

// exclusive prefix sum of n_tracks over the frames (one workgroup; F <= a few 100 k)
__global__ void __launch_bounds__(1024)
row_offsets_kernel(const int32_t* __restrict__ n_tracks, int F, int T, int32_t* __restrict__ offsets, int32_t* __restrict__ total) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (F + 1023) / 1024;
    const int lo = tid * per, hi = min(F, lo + per);
    int s = 0;
    for (int f = lo; f < hi; ++f) { const int n = n_tracks[f]; s += n < 0 ? 0 : (n > T ? T : n); }
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;
    for (int f = lo; f < hi; ++f) { offsets[f] = run; const int n = n_tracks[f]; run += n < 0 ? 0 : (n > T ? T : n); }
    if (tid == 1023) *total = part[1023];
}

// an empty shard (world > number of chains): the full header with zero chains and an empty local section, so that the stitch's
// layout check passes
__global__ void empty_header_kernel(int L, int T, int row_cap, MsgLayout lay, int32_t* __restrict__ h) {
    if (threadIdx.x == 0) { h[0] = 0; h[1] = L; h[2] = T; h[3] = 0; h[4] = 0; h[5] = row_cap; h[6] = 0; h[7] = 0; }
    if (threadIdx.x < ST_LOCAL_HDR) h[lay.off_local() + threadIdx.x] = 0;
}

// one wave per frame.  The tables have TT slots per frame (the chain kernel's t_max, or the repair tier's wider tables); the message
// has T >= the live tracklets of any frame -- the same T on every rank, whatever a rank's tables look like.
__global__ void __launch_bounds__(64)
pack_kernel(const double* __restrict__ params, const double* __restrict__ joints, const int32_t* __restrict__ meta,
            const int32_t* __restrict__ n_tracks, const int32_t* __restrict__ next_id, const int32_t* __restrict__ offsets,
            const int32_t* __restrict__ total, const uint32_t* __restrict__ void_words, int n_void_words, int F, int L, int TT, int T,
            MsgLayout lay, uint32_t* __restrict__ msg) {
    const int f = blockIdx.x, lane = threadIdx.x;
    int nt = n_tracks[f];
    nt = nt < 0 ? 0 : (nt > T ? T : nt);
    nt = nt > TT ? TT : nt;
    const int b = f / L, t = f - b * L;
    if (f == 0 && lane == 0) {
        int32_t* h = reinterpret_cast<int32_t*>(msg);
        h[0] = F / L; h[1] = L; h[2] = T; h[3] = min(*total, lay.row_cap); h[4] = *total; h[5] = lay.row_cap; h[6] = F; h[7] = 0;
        // the local section's header: {local roots, matched pairs, error word, void word}; the first three are the local stitch's
        int32_t* lh = h + lay.off_local();
        uint32_t vw = 0;
        for (int i = 0; i < n_void_words; ++i) vw |= void_words[i] ? (1u << i) : 0u;
        lh[0] = 0; lh[1] = 0; lh[2] = 0; lh[3] = (int32_t)vw; lh[4] = 0; lh[5] = 0; lh[6] = 0; lh[7] = 0;
    }
    if (t == 0 && lane == 0) reinterpret_cast<int32_t*>(msg + lay.off_ids())[b] = next_id[b];
    const int row0 = offsets[f];
    for (int s = 0; s < nt; ++s) {
        const int row = row0 + s;
        if (row >= lay.row_cap) break;
        uint32_t* dst = msg + lay.off_rows() + (size_t)row * MVMC_ROW_WORDS;
        const int32_t* m = meta + ((size_t)f * TT + s) * 4;
        if (lane < 6) {
            const int32_t v = lane == 0 ? f : (lane == 1 ? s : m[lane - 2]);
            dst[lane] = (uint32_t)v;
        }
        if (lane < 54) dst[6 + lane] = __float_as_uint((float)joints[((size_t)f * TT + s) * 54 + lane]);
        for (int e = lane; e < 68; e += 64) dst[60 + e] = __float_as_uint((float)params[((size_t)f * TT + s) * 68 + e]);
    }
    // first / last frame table of the chain (a chain of one frame writes both)
    for (int side = 0; side < 2; ++side) {
        if (t != (side == 0 ? 0 : L - 1)) continue;
        uint32_t* bt = msg + lay.off_bounds() + ((size_t)b * 2 + side) * T * MVMC_BOUND_WORDS;
        for (int s = 0; s < T; ++s) {
            uint32_t* dst = bt + (size_t)s * MVMC_BOUND_WORDS;
            const bool live = s < nt;
            if (lane == 0) dst[0] = (uint32_t)(live ? meta[((size_t)f * TT + s) * 4] : -1);
            if (lane < 54) dst[1 + lane] = live ? __float_as_uint((float)joints[((size_t)f * TT + s) * 54 + lane]) : 0x7fc00000u;
            if (lane == 54) dst[55] = 0u;
        }
    }
}

// Optimal assignment of n rows to m >= n columns (Kuhn-Munkres with potentials, O(n^2 m)); col_of[i] = column of row i.
// Costs must be finite (the caller replaces non-finite entries by a large constant): with a NaN row no column is ever selected and
// the augmenting loop would never end.  Every loop is bounded regardless (an augmenting path visits a column at most once, and the
// back-trace has at most m links); false = the bound was hit, col_of is then not a valid assignment.
template <typename Mat>
__device__ bool assign_rows(const Mat& a, int n, int m, int* col_of) {
    double u[ST_T + 1], v[ST_T + 1], minv[ST_T + 1];
    int p[ST_T + 1], way[ST_T + 1];
    bool used[ST_T + 1];
    for (int j = 0; j <= m; ++j) { v[j] = 0.0; p[j] = 0; way[j] = 0; }
    for (int i = 0; i <= n; ++i) u[i] = 0.0;
    for (int i = 1; i <= n; ++i) {
        p[0] = i;
        int j0 = 0, rounds = 0;
        for (int j = 0; j <= m; ++j) { minv[j] = 1e300; used[j] = false; }
        do {
            used[j0] = true;
            const int i0 = p[j0];
            double delta = 1e300;
            int j1 = 0;
            for (int j = 1; j <= m; ++j)
                if (!used[j]) {
                    const double cur = a[i0 - 1][j - 1] - u[i0] - v[j];
                    if (cur < minv[j]) { minv[j] = cur; way[j] = j0; }
                    if (minv[j] < delta) { delta = minv[j]; j1 = j; }
                }
            if (j1 == 0 || ++rounds > m + 1) return false;
            for (int j = 0; j <= m; ++j)
                if (used[j]) { u[p[j]] += delta; v[j] -= delta; } else minv[j] -= delta;
            j0 = j1;
        } while (p[j0] != 0);
        int links = 0;
        do { const int j1 = way[j0]; p[j0] = p[j1]; j0 = j1; if (++links > m + 1) return false; } while (j0);
    }
    for (int i = 0; i < n; ++i) col_of[i] = -1;
    for (int j = 1; j <= m; ++j)
        if (p[j] > 0) col_of[p[j] - 1] = j - 1;
    for (int i = 0; i < n; ++i)
        if (col_of[i] < 0) return false;
    return true;
}

// One chain boundary on one wave: the live tracklets of `pv` (last frame of the earlier chain) against those of `nx` (first frame of
// the later one), T slots of MVMC_BOUND_WORDS each.  Lane q computes the costs of the pairs q, q + 64, ...; lane 0 solves the
// assignment.  match_row[s] (T, pre-set to -1 by the caller) = slot of pv matched to slot s of nx; link[l] (IC, pre-set by the caller)
// receives, for the local id l of a matched tracklet of nx, the local id of its partner in pv.  Returns (on lane 0) the number of
// pairs, or -1 if the assignment did not terminate.
struct BoundScratch {
    double cost[ST_T][ST_T];
    int ip[ST_T], in[ST_T];
};
__device__ inline int match_boundary_wave(const uint32_t* __restrict__ pv, const uint32_t* __restrict__ nx, int T, int IC,
                                          double max_dist, BoundScratch& B, int32_t* __restrict__ match_row,
                                          int32_t* __restrict__ link) {
    const int lane = threadIdx.x & 63;
    const bool lp = lane < T && (int32_t)pv[(size_t)lane * MVMC_BOUND_WORDS] >= 0;
    const bool ln = lane < T && (int32_t)nx[(size_t)lane * MVMC_BOUND_WORDS] >= 0;
    const unsigned long long mp = __builtin_amdgcn_ballot_w64(lp), mn = __builtin_amdgcn_ballot_w64(ln);
    const int np = __popcll(mp), nn = __popcll(mn);
    if (lp) B.ip[__popcll(mp & ((1ull << lane) - 1ull))] = lane;
    if (ln) B.in[__popcll(mn & ((1ull << lane) - 1ull))] = lane;
    __syncthreads();
    if (np == 0 || nn == 0) return 0;
    const bool swap = np > nn;          // rows = the smaller side
    const int nr = swap ? nn : np, nc = swap ? np : nn;
    for (int q = lane; q < np * nn; q += 64) {
        const int i = q / nn, j = q - i * nn;
        const uint32_t* a = pv + (size_t)B.ip[i] * MVMC_BOUND_WORDS + 1;
        const uint32_t* b = nx + (size_t)B.in[j] * MVMC_BOUND_WORDS + 1;
        double sum = 0.0;
        for (int k = 0; k < 18; ++k) {
            const double dx = (double)__uint_as_float(a[3 * k]) - (double)__uint_as_float(b[3 * k]);
            const double dy = (double)__uint_as_float(a[3 * k + 1]) - (double)__uint_as_float(b[3 * k + 1]);
            const double dz = (double)__uint_as_float(a[3 * k + 2]) - (double)__uint_as_float(b[3 * k + 2]);
            sum += sqrt(dx * dx + dy * dy + dz * dz);
        }
        // a tracklet with a non-finite joint matches nobody: a cost beyond any max_dist, and finite for the assignment
        const double c = isfinite(sum) ? sum / 18.0 : MVMC_STITCH_NO_MATCH;
        if (swap) B.cost[j][i] = c; else B.cost[i][j] = c;
    }
    __syncthreads();
    int pairs = 0;
    if (lane == 0) {
        int col_of[ST_T];
        if (!assign_rows(B.cost, nr, nc, col_of)) return -1;
        for (int r = 0; r < nr; ++r) {
            const int i = swap ? col_of[r] : r, j = swap ? r : col_of[r];
            if (!(B.cost[r][col_of[r]] <= max_dist)) continue;
            match_row[B.in[j]] = B.ip[i];
            const int lid_n = (int32_t)nx[(size_t)B.in[j] * MVMC_BOUND_WORDS], lid_p = (int32_t)pv[(size_t)B.ip[i] * MVMC_BOUND_WORDS];
            if (lid_n < IC && lid_p < IC) link[lid_n] = lid_p;
            ++pairs;
        }
    }
    return pairs;
}

// ---- before the gather: the shard's own boundaries (a wave per chain) ----
__global__ void __launch_bounds__(64)
local_match_kernel(uint32_t* __restrict__ msg, MsgLayout lay, int B, double max_dist, int32_t* __restrict__ ptr) {
    __shared__ BoundScratch S;
    __shared__ int32_t link[64];
    const int b = blockIdx.x, lane = threadIdx.x, T = lay.t_max, IC = lay.id_cap;
    int32_t* lh = reinterpret_cast<int32_t*>(msg + lay.off_local());
    int32_t* lmatch = reinterpret_cast<int32_t*>(msg + lay.off_lmatch()) + (size_t)b * T;
    const int n_ids = reinterpret_cast<const int32_t*>(msg + lay.off_ids())[b];
    if (lane == 0 && n_ids > IC) atomicOr(lh + 2, 1);
    if (lane < T) lmatch[lane] = -1;
    if (lane < IC) link[lane] = -1;
    __syncthreads();
    int pairs = 0;
    if (b > 0) {
        const uint32_t* pv = msg + lay.off_bounds() + ((size_t)(b - 1) * 2 + 1) * T * MVMC_BOUND_WORDS;
        const uint32_t* nx = msg + lay.off_bounds() + ((size_t)b * 2 + 0) * T * MVMC_BOUND_WORDS;
        pairs = match_boundary_wave(pv, nx, T, IC, max_dist, S, lmatch, link);
        if (lane == 0 && pairs < 0) atomicOr(lh + 2, 2);
        if (lane == 0 && pairs > 0) atomicAdd(lh + 1, pairs);
    }
    __syncthreads();
    // node (b, l) points at its predecessor in chain b - 1, at itself if it has none, nowhere if the chain has no such id
    if (lane < IC) ptr[(size_t)b * IC + lane] = lane < n_ids ? (link[lane] >= 0 ? (b - 1) * IC + link[lane] : b * IC + lane) : -1;
}

// roots numbered in chain order, pointer jumping, local identities -> the message (one workgroup)
__global__ void __launch_bounds__(1024)
local_ids_kernel(uint32_t* __restrict__ msg, MsgLayout lay, int B, int32_t* __restrict__ ptr, int32_t* __restrict__ rank_of) {
    __shared__ int part[1024];
    const int tid = threadIdx.x, IC = lay.id_cap;
    const int n_nodes = B * IC;
    const int per = (n_nodes + 1023) / 1024;
    const int lo = min(n_nodes, tid * per), hi = min(n_nodes, lo + per);
    int s = 0;
    for (int k = lo; k < hi; ++k) s += ptr[k] == k;
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;
    for (int k = lo; k < hi; ++k) rank_of[k] = ptr[k] == k ? run++ : -1;
    __syncthreads();
    // a tracklet's chain of predecessors has at most B links
    for (int span = 1; span < B; span <<= 1) {
        for (int k = tid; k < n_nodes; k += 1024) {
            const int p = ptr[k];
            if (p >= 0) { const int q = ptr[p]; if (q != p) ptr[k] = q; }
        }
        __syncthreads();
    }
    int32_t* lgid = reinterpret_cast<int32_t*>(msg + lay.off_lgid());
    for (int k = tid; k < n_nodes; k += 1024) {
        const int p = ptr[k];
        lgid[k] = p >= 0 ? rank_of[p] : -1;
    }
    if (tid == 0) reinterpret_cast<int32_t*>(msg + lay.off_local())[0] = part[1023];
}

// ---- after the gather ----
struct StitchArgs {
    const uint32_t* msgs;     // world messages, msg_words apart
    size_t msg_words;
    int world, id_cap;
    MsgLayout lay;
    double max_dist;
    int32_t* gid;             // (Btot, id_cap) out: global identity of (chain, local id), -1 = no such local id
    int32_t* match;           // (Btot, T) out: slot of the previous chain's last frame matched to slot s of this chain's first frame, -1
    int32_t* info;            // (4) out: {Btot, number of global identities, error word, matched pairs}
    // workspace, per shard r: xlink (id_cap): local identity IN THE PREVIOUS NON-EMPTY SHARD that local id l of the shard's first chain
    // continues, or -1; xmatch (T): the boundary's match row; resolved (id_cap): global identity of a linked l; then
    // {previous non-empty shard or -1, base, link mask lo, link mask hi, ids of the first chain, pairs, 0, 0}
    int32_t* work;
    __host__ __device__ int stride() const { return 2 * id_cap + lay.t_max + 8; }
    __device__ int32_t* xlink(int r) const { return work + (size_t)r * stride(); }
    __device__ int32_t* xmatch(int r) const { return xlink(r) + id_cap; }
    __device__ int32_t* resolved(int r) const { return xmatch(r) + lay.t_max; }
    __device__ int32_t* shard(int r) const { return resolved(r) + id_cap; }
    __device__ const uint32_t* msg(int r) const { return msgs + (size_t)r * msg_words; }
    __device__ const int32_t* hdr(int r) const { return reinterpret_cast<const int32_t*>(msg(r)); }
};

// the shard boundaries: block r matches the first chain of shard r against the last chain of the nearest non-empty shard before it
__global__ void __launch_bounds__(64)
shard_bound_kernel(StitchArgs A) {
    __shared__ BoundScratch S;
    __shared__ int32_t link[64];
    const int r = blockIdx.x, lane = threadIdx.x, T = A.lay.t_max, IC = A.id_cap;
    int32_t* sh = A.shard(r);
    if (lane < IC) { A.xlink(r)[lane] = -1; link[lane] = -1; }
    if (lane < T) A.xmatch(r)[lane] = -1;
    if (lane < 8) sh[lane] = lane == 0 ? -1 : 0;
    __syncthreads();
    const int Br = A.hdr(r)[0];
    if (Br <= 0 || Br > A.lay.b_cap) return;
    int rp = r - 1;
    while (rp >= 0 && A.hdr(rp)[0] <= 0) --rp;
    int n_first = reinterpret_cast<const int32_t*>(A.msg(r) + A.lay.off_ids())[0];
    n_first = n_first > IC ? IC : n_first;
    if (lane == 0) { sh[0] = rp; sh[4] = n_first; }
    if (rp < 0 || A.hdr(rp)[0] > A.lay.b_cap) return;
    const int Bp = A.hdr(rp)[0];
    const uint32_t* pv = A.msg(rp) + A.lay.off_bounds() + ((size_t)(Bp - 1) * 2 + 1) * T * MVMC_BOUND_WORDS;
    const uint32_t* nx = A.msg(r) + A.lay.off_bounds() + (size_t)0 * T * MVMC_BOUND_WORDS;
    const int pairs = match_boundary_wave(pv, nx, T, IC, A.max_dist, S, A.xmatch(r), link);
    __syncthreads();
    if (lane == 0) {
        if (pairs < 0) atomicOr(A.info + 2, 2);
        sh[5] = pairs > 0 ? pairs : 0;
    }
    // the partner's local id in the last chain of shard rp -> its shard-local identity
    const int32_t* lgid_p = reinterpret_cast<const int32_t*>(A.msg(rp) + A.lay.off_lgid()) + (size_t)(Bp - 1) * IC;
    bool linked = false;
    if (lane < n_first && link[lane] >= 0) {
        const int k = lgid_p[link[lane]];
        if (k >= 0) { A.xlink(r)[lane] = k; linked = true; }
    }
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(linked);
    if (lane == 0) { sh[2] = (int32_t)(uint32_t)mask; sh[3] = (int32_t)(uint32_t)(mask >> 32); }
}

// global identity of shard-local identity k of shard r (tables of shard r complete)
__device__ __forceinline__ int global_of(const StitchArgs& A, int r, int k) {
    const int32_t* sh = A.shard(r);
    const unsigned long long mask = (unsigned long long)(uint32_t)sh[2] | ((unsigned long long)(uint32_t)sh[3] << 32);
    const int n_first = sh[4];
    if (k < n_first && ((mask >> k) & 1ull)) return A.resolved(r)[k];
    const int below = k < 64 ? __popcll(mask & ((1ull << k) - 1ull)) : __popcll(mask);
    return sh[1] + k - below;
}

// local -> global numbers, shard after shard (one thread: at most world x id_cap steps), and the layout checks
__global__ void __launch_bounds__(64)
shard_resolve_kernel(StitchArgs A) {
    if (threadIdx.x != 0) return;
    int base = 0, chains = 0, pairs = 0, flag = 0;
    for (int r = 0; r < A.world; ++r) {
        const int32_t* h = A.hdr(r);
        if (h[4] > h[5] || h[0] > A.lay.b_cap || h[2] != A.lay.t_max || h[0] < 0) { flag |= 1; continue; }   // rows dropped / layout mismatch
        const int32_t* lh = h + A.lay.off_local();
        int32_t* sh = A.shard(r);
        sh[1] = base;
        if (h[0] == 0) continue;
        flag |= lh[2] | (lh[3] ? 4 : 0);
        const unsigned long long mask = (unsigned long long)(uint32_t)sh[2] | ((unsigned long long)(uint32_t)sh[3] << 32);
        const int rp = sh[0];
        for (int l = 0; l < sh[4]; ++l)
            if ((mask >> l) & 1ull) A.resolved(r)[l] = global_of(A, rp, A.xlink(r)[l]);
        base += lh[0] - __popcll(mask);
        chains += h[0];
        pairs += lh[1] + sh[5];
    }
    A.info[0] = chains; A.info[1] = base; A.info[3] = pairs;
    if (flag) atomicOr(A.info + 2, flag);
}

// the tables: one thread per (chain, local id | slot)
__global__ void __launch_bounds__(256)
stitch_expand_kernel(StitchArgs A) {
    __shared__ int s_first[64 + 1];
    const int T = A.lay.t_max, IC = A.id_cap, W = IC > T ? IC : T;
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int r = 0; r < A.world; ++r) {
            s_first[r] = acc;
            const int n = A.hdr(r)[0];
            acc += (n > 0 && n <= A.lay.b_cap) ? n : 0;
        }
        s_first[A.world] = acc;
    }
    __syncthreads();
    const int Btot = s_first[A.world];
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const int g = (int)(e / W), l = (int)(e - (long long)g * W);
    if (g >= Btot) return;
    int r = 0;
    while (r + 1 < A.world && s_first[r + 1] <= g) ++r;
    const int b = g - s_first[r];
    if (l < IC) {
        const int k = reinterpret_cast<const int32_t*>(A.msg(r) + A.lay.off_lgid())[(size_t)b * IC + l];
        A.gid[(size_t)g * IC + l] = k >= 0 ? global_of(A, r, k) : -1;
    }
    if (l < T)
        A.match[(size_t)g * T + l] = b > 0 ? reinterpret_cast<const int32_t*>(A.msg(r) + A.lay.off_lmatch())[(size_t)b * T + l] : A.xmatch(r)[l];
}

}  // namespace

extern "C" long long mvmc_pack_message_words(int n_chains_cap, int t_max, int row_cap, int id_cap) {
    if (n_chains_cap < 0 || t_max <= 0 || t_max > ST_T || row_cap < 0 || id_cap <= 0 || id_cap > 64) return -1;
    MsgLayout lay{n_chains_cap, t_max, row_cap, id_cap};
    return (long long)lay.words();
}

extern "C" long long mvmc_pack_work_words(int n_frames, int chain_len, int id_cap) {
    if (n_frames < 0 || chain_len <= 0 || id_cap <= 0 || id_cap > 64) return -1;
    return (long long)n_frames + 1 + 2LL * (n_frames / chain_len) * id_cap;
}

extern "C" int mvmc_pack_tracks(const double* out_params, const double* out_joints, const int32_t* out_meta,
                                const int32_t* out_n_tracks, const int32_t* next_id, int n_frames, int chain_len, int t_tables,
                                int t_max, int n_chains_cap, int row_cap, int id_cap, double max_dist, const uint32_t* void_words,
                                int n_void_words, int32_t* work, void* message, mvmcStream_t stream) {
    if (!out_params || !out_joints || !out_meta || !out_n_tracks || !next_id || !work || !message) return MVMC_ERR_ARG;
    if (n_frames < 0 || chain_len <= 0 || n_frames % chain_len || t_max <= 0 || t_max > ST_T || t_tables <= 0 || row_cap < 0)
        return MVMC_ERR_ARG;
    if (id_cap <= 0 || id_cap > 64 || n_void_words < 0 || n_void_words > 32 || (n_void_words && !void_words)) return MVMC_ERR_ARG;
    if (n_frames / chain_len > n_chains_cap) return MVMC_ERR_ARG;
    MsgLayout lay{n_chains_cap, t_max, row_cap, id_cap};
    hipStream_t s = (hipStream_t)stream;
    if (n_frames == 0) {
        hipLaunchKernelGGL(empty_header_kernel, dim3(1), dim3(64), 0, s, chain_len, t_max, row_cap, lay, (int32_t*)message);
        MVMC_CHECK_LAUNCH();
        return MVMC_OK;
    }
    const int B = n_frames / chain_len;
    // work: row offsets (n_frames + 1 words, the last one receives the total), then the local stitch's two node arrays
    int32_t* row_offsets = work;
    int32_t* ptr = work + n_frames + 1;
    int32_t* rank_of = ptr + (size_t)B * id_cap;
    hipLaunchKernelGGL(row_offsets_kernel, dim3(1), dim3(1024), 0, s, out_n_tracks, n_frames, t_max < t_tables ? t_max : t_tables, row_offsets,
                       row_offsets + n_frames);
    hipLaunchKernelGGL(pack_kernel, dim3(n_frames), dim3(64), 0, s, out_params, out_joints, out_meta, out_n_tracks, next_id, row_offsets,
                       row_offsets + n_frames, void_words, n_void_words, n_frames, chain_len, t_tables, t_max, lay, (uint32_t*)message);
    hipLaunchKernelGGL(local_match_kernel, dim3(B), dim3(64), 0, s, (uint32_t*)message, lay, B, max_dist, ptr);
    hipLaunchKernelGGL(local_ids_kernel, dim3(1), dim3(1024), 0, s, (uint32_t*)message, lay, B, ptr, rank_of);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}

extern "C" long long mvmc_stitch_work_words(int world, int t_max, int id_cap) {
    if (world <= 0 || world > 64 || t_max <= 0 || t_max > ST_T || id_cap <= 0 || id_cap > 64) return -1;
    return (long long)world * (2 * id_cap + t_max + 8);
}

extern "C" int mvmc_stitch_chains(const void* messages, long long message_words, int world, int n_chains_cap, int t_max, int row_cap,
                                  int id_cap, double max_dist, int n_chains_total_cap, int32_t* gid, int32_t* match, int32_t* info,
                                  int32_t* work, mvmcStream_t stream) {
    if (!messages || !gid || !match || !info || !work) return MVMC_ERR_ARG;
    if (world <= 0 || world > 64 || t_max <= 0 || t_max > ST_T || id_cap <= 0 || id_cap > 64 || n_chains_total_cap < 0) return MVMC_ERR_ARG;
    MsgLayout lay{n_chains_cap, t_max, row_cap, id_cap};
    if (message_words < (long long)lay.words()) return MVMC_ERR_ARG;
    if ((long long)world * n_chains_cap > n_chains_total_cap) return MVMC_ERR_ARG;
    StitchArgs A;
    A.msgs = (const uint32_t*)messages; A.msg_words = (size_t)message_words; A.world = world; A.id_cap = id_cap; A.lay = lay;
    A.max_dist = max_dist; A.gid = gid; A.match = match; A.info = info; A.work = work;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(info, 0, 4 * sizeof(int32_t), s) != hipSuccess) return MVMC_ERR_LAUNCH;
    hipLaunchKernelGGL(shard_bound_kernel, dim3(world), dim3(64), 0, s, A);
    hipLaunchKernelGGL(shard_resolve_kernel, dim3(1), dim3(64), 0, s, A);
    const long long cap = (long long)world * n_chains_cap, W = id_cap > t_max ? id_cap : t_max;
    if (cap > 0) hipLaunchKernelGGL(stitch_expand_kernel, dim3((unsigned)((cap * W + 255) / 256)), dim3(256), 0, s, A);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}
