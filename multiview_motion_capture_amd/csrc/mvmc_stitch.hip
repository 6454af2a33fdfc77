// Multi-GPU glue of the frame-sharded path.  The reference has no counterpart: its tracker is one
// sequential pass over the sequence (motion_capture.py:1062-1116).  Here a sequence is cut into chains (sub-sequences that cold-start,
// DESIGN.md section 7), contiguous chain ranges go to the GPUs, and after ONE all-gather of the packed results every rank stitches
// the identities across all chain boundaries -- shard boundaries are chain boundaries like any other.
//
//   pack_tracks   a shard's per-frame tracklet tables (mvmc_chain_run's outputs: (F,T,...) padded to T slots) -> one message:
//                 only live tracklets, float32, plus the first / last frame table of every chain for the stitch
//   stitch        gathered messages -> for every chain boundary the optimal assignment (mean joint distance, pairs farther than
//                 max_dist dropped), then global identities by pointer jumping along the matched tracklets
#include "mvmc_common.h"

namespace {

constexpr int ST_T = 16;        // max tracklet slots per frame (t_max <= 16)
constexpr int ST_HDR = 8;       // header words
constexpr double MVMC_STITCH_NO_MATCH = 1e30;

struct MsgLayout {
    int b_cap, t_max, row_cap;
    __host__ __device__ size_t off_ids() const { return ST_HDR; }
    __host__ __device__ size_t off_bounds() const { return off_ids() + (size_t)b_cap; }
    __host__ __device__ size_t off_rows() const { return off_bounds() + (size_t)b_cap * 2 * t_max * MVMC_BOUND_WORDS; }
    __host__ __device__ size_t words() const { return off_rows() + (size_t)row_cap * MVMC_ROW_WORDS; }
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

// an empty shard (world > number of chains): the full header with zero chains, so that the stitch's layout check passes
__global__ void empty_header_kernel(int L, int T, int row_cap, int32_t* __restrict__ h) {
    if (threadIdx.x == 0) { h[0] = 0; h[1] = L; h[2] = T; h[3] = 0; h[4] = 0; h[5] = row_cap; h[6] = 0; h[7] = 0; }
}

// one wave per frame
__global__ void __launch_bounds__(64)
pack_kernel(const double* __restrict__ params, const double* __restrict__ joints, const int32_t* __restrict__ meta,
            const int32_t* __restrict__ n_tracks, const int32_t* __restrict__ next_id, const int32_t* __restrict__ offsets,
            const int32_t* __restrict__ total, int F, int L, int T, MsgLayout lay, uint32_t* __restrict__ msg) {
    const int f = blockIdx.x, lane = threadIdx.x;
    int nt = n_tracks[f];
    nt = nt < 0 ? 0 : (nt > T ? T : nt);
    const int b = f / L, t = f - b * L;
    if (f == 0 && lane == 0) {
        int32_t* h = reinterpret_cast<int32_t*>(msg);
        h[0] = F / L; h[1] = L; h[2] = T; h[3] = min(*total, lay.row_cap); h[4] = *total; h[5] = lay.row_cap; h[6] = F; h[7] = 0;
    }
    if (t == 0 && lane == 0) reinterpret_cast<int32_t*>(msg + lay.off_ids())[b] = next_id[b];
    const int row0 = offsets[f];
    for (int s = 0; s < nt; ++s) {
        const int row = row0 + s;
        if (row >= lay.row_cap) break;
        uint32_t* dst = msg + lay.off_rows() + (size_t)row * MVMC_ROW_WORDS;
        const int32_t* m = meta + ((size_t)f * T + s) * 4;
        if (lane < 6) {
            const int32_t v = lane == 0 ? f : (lane == 1 ? s : m[lane - 2]);
            dst[lane] = (uint32_t)v;
        }
        if (lane < 54) dst[6 + lane] = __float_as_uint((float)joints[((size_t)f * T + s) * 54 + lane]);
        for (int e = lane; e < 68; e += 64) dst[60 + e] = __float_as_uint((float)params[((size_t)f * T + s) * 68 + e]);
    }
    // first / last frame table of the chain (a chain of one frame writes both)
    for (int side = 0; side < 2; ++side) {
        if (t != (side == 0 ? 0 : L - 1)) continue;
        uint32_t* bt = msg + lay.off_bounds() + ((size_t)b * 2 + side) * T * MVMC_BOUND_WORDS;
        for (int s = 0; s < T; ++s) {
            uint32_t* dst = bt + (size_t)s * MVMC_BOUND_WORDS;
            const bool live = s < nt;
            if (lane == 0) dst[0] = (uint32_t)(live ? meta[((size_t)f * T + s) * 4] : -1);
            if (lane < 54) dst[1 + lane] = live ? __float_as_uint((float)joints[((size_t)f * T + s) * 54 + lane]) : 0x7fc00000u;
            if (lane == 54) dst[55] = 0u;
        }
    }
}

// Optimal assignment of n rows to m >= n columns (Kuhn-Munkres with potentials, O(n^2 m)); col_of[i] = column of row i.
// Costs must be finite (the caller replaces non-finite entries by a large constant): with a NaN row no column is ever selected and
// the augmenting loop would never end.  Every loop is bounded regardless (an augmenting path visits a column at most once, and the
// back-trace has at most m links); false = the bound was hit, col_of is then not a valid assignment.
__device__ bool assign_rows(const double (&a)[ST_T][ST_T], int n, int m, int* col_of) {
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

struct StitchArgs {
    const uint32_t* msgs;     // world messages, msg_words apart
    size_t msg_words;
    int world, id_cap;
    MsgLayout lay;
    double max_dist;
    int32_t* gid;             // (Btot, id_cap) out: global identity of (chain, local id), -1 = no such local id
    int32_t* match;           // (Btot, T) out: slot of the previous chain's last frame matched to slot s of this chain's first frame, -1
    int32_t* info;            // (4) out: {Btot, number of global identities, overflow flag, matched pairs}
    int32_t* ptr;             // (Btot * id_cap) workspace
    int32_t* rank_of;         // (Btot * id_cap) workspace
};

// first global chain of every rank (world <= 64) and the layout checks; every thread computes the same
__device__ __forceinline__ int chain_starts(const StitchArgs& A, int* first, int* flag) {
    int acc = 0, fl = 0;
    for (int r = 0; r < A.world; ++r) {
        const int32_t* h = reinterpret_cast<const int32_t*>(A.msgs + (size_t)r * A.msg_words);
        first[r] = acc;
        acc += h[0];
        if (h[4] > h[5] || h[0] > A.lay.b_cap || h[2] != A.lay.t_max) fl = 1;   // rows dropped / layout mismatch
    }
    first[A.world] = acc;
    *flag = fl;
    return acc;
}

// phase A: one thread per chain boundary (many workgroups: the assignments are independent)
__global__ void __launch_bounds__(64)
stitch_match_kernel(StitchArgs A) {
    __shared__ int s_first[64 + 1];
    __shared__ int s_flag;
    const int T = A.lay.t_max, IC = A.id_cap;
    if (threadIdx.x == 0) { int fl; chain_starts(A, s_first, &fl); s_flag = fl; }
    __syncthreads();
    const int Btot = s_first[A.world];
    auto chain_msg = [&](int g, int* b_local) {
        int r = 0;
        while (r + 1 < A.world && s_first[r + 1] <= g) ++r;
        *b_local = g - s_first[r];
        return A.msgs + (size_t)r * A.msg_words;
    };
    const int g = blockIdx.x * 64 + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x == 0 && s_flag) atomicOr(A.info + 2, 1);
    if (g >= Btot) return;
    int bl;
    const uint32_t* mn = chain_msg(g, &bl);
    const int n_ids = reinterpret_cast<const int32_t*>(mn + A.lay.off_ids())[bl];
    if (n_ids > IC) atomicOr(A.info + 2, 1);
    const uint32_t* nx = mn + A.lay.off_bounds() + ((size_t)bl * 2 + 0) * T * MVMC_BOUND_WORDS;
    for (int l = 0; l < IC; ++l) A.ptr[(size_t)g * IC + l] = l < n_ids ? g * IC + l : -1;
    for (int s = 0; s < T; ++s) A.match[(size_t)g * T + s] = -1;
    if (g == 0) return;
    int bp;
    const uint32_t* mp = chain_msg(g - 1, &bp);
    const uint32_t* pv = mp + A.lay.off_bounds() + ((size_t)bp * 2 + 1) * T * MVMC_BOUND_WORDS;
    int ip[ST_T], in[ST_T], np = 0, nn = 0;
    for (int s = 0; s < T; ++s) {
        if ((int32_t)pv[(size_t)s * MVMC_BOUND_WORDS] >= 0) ip[np++] = s;
        if ((int32_t)nx[(size_t)s * MVMC_BOUND_WORDS] >= 0) in[nn++] = s;
    }
    if (np == 0 || nn == 0) return;
    double cost[ST_T][ST_T];   // rows = the smaller side
    const bool swap = np > nn;
    const int nr = swap ? nn : np, nc = swap ? np : nn;
    for (int i = 0; i < np; ++i)
        for (int j = 0; j < nn; ++j) {
            const uint32_t* a = pv + (size_t)ip[i] * MVMC_BOUND_WORDS + 1;
            const uint32_t* b = nx + (size_t)in[j] * MVMC_BOUND_WORDS + 1;
            double sum = 0.0;
            for (int k = 0; k < 18; ++k) {
                const double dx = (double)__uint_as_float(a[3 * k]) - (double)__uint_as_float(b[3 * k]);
                const double dy = (double)__uint_as_float(a[3 * k + 1]) - (double)__uint_as_float(b[3 * k + 1]);
                const double dz = (double)__uint_as_float(a[3 * k + 2]) - (double)__uint_as_float(b[3 * k + 2]);
                sum += sqrt(dx * dx + dy * dy + dz * dz);
            }
            // a tracklet with a non-finite joint matches nobody: a cost beyond any max_dist, and finite for the assignment
            const double c = isfinite(sum) ? sum / 18.0 : MVMC_STITCH_NO_MATCH;
            if (swap) cost[j][i] = c; else cost[i][j] = c;
        }
    int col_of[ST_T];
    if (!assign_rows(cost, nr, nc, col_of)) { atomicOr(A.info + 2, 2); return; }   // bit 1: an assignment did not terminate
    int pairs = 0;
    for (int r = 0; r < nr; ++r) {
        const int i = swap ? col_of[r] : r, j = swap ? r : col_of[r];
        if (!(cost[r][col_of[r]] <= A.max_dist)) continue;
        A.match[(size_t)g * T + in[j]] = ip[i];
        const int lid_n = (int32_t)nx[(size_t)in[j] * MVMC_BOUND_WORDS], lid_p = (int32_t)pv[(size_t)ip[i] * MVMC_BOUND_WORDS];
        if (lid_n < IC && lid_p < IC) A.ptr[(size_t)g * IC + lid_n] = (g - 1) * IC + lid_p;
        ++pairs;
    }
    if (pairs) atomicAdd(A.info + 3, pairs);
}

// phases B-D on one workgroup: roots numbered in chain order, pointer jumping, global identities
__global__ void __launch_bounds__(1024)
stitch_ids_kernel(StitchArgs A) {
    __shared__ int s_first[64 + 1];
    __shared__ int part[1024];
    const int tid = threadIdx.x, IC = A.id_cap;
    if (tid == 0) { int fl; chain_starts(A, s_first, &fl); }
    __syncthreads();
    const int Btot = s_first[A.world];
    // ---- phase B: roots (tracklets without a predecessor) numbered in chain order ----
    const int n_nodes = Btot * IC;
    const int per = (n_nodes + 1023) / 1024;
    const int lo = tid * per, hi = min(n_nodes, lo + per);
    int s = 0;
    for (int k = lo; k < hi; ++k) s += A.ptr[k] == k;
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;
    for (int k = lo; k < hi; ++k) A.rank_of[k] = A.ptr[k] == k ? run++ : -1;
    __syncthreads();
    // ---- phase C: pointer jumping (a tracklet's chain of predecessors has at most Btot links) ----
    for (int span = 1; span < Btot; span <<= 1) {
        for (int k = tid; k < n_nodes; k += 1024) {
            const int p = A.ptr[k];
            if (p >= 0) { const int q = A.ptr[p]; if (q != p) A.ptr[k] = q; }
        }
        __syncthreads();
    }
    // ---- phase D ----
    for (int k = tid; k < n_nodes; k += 1024) {
        const int p = A.ptr[k];
        A.gid[k] = p >= 0 ? A.rank_of[p] : -1;
    }
    if (tid == 0) { A.info[0] = Btot; A.info[1] = part[1023]; }
}

}  // namespace

extern "C" long long mvmc_pack_message_words(int n_chains_cap, int t_max, int row_cap) {
    if (n_chains_cap < 0 || t_max <= 0 || t_max > ST_T || row_cap < 0) return -1;
    MsgLayout lay{n_chains_cap, t_max, row_cap};
    return (long long)lay.words();
}

extern "C" int mvmc_pack_tracks(const double* out_params, const double* out_joints, const int32_t* out_meta,
                                const int32_t* out_n_tracks, const int32_t* next_id, int n_frames, int chain_len, int t_max,
                                int n_chains_cap, int row_cap, int32_t* row_offsets, void* message, mvmcStream_t stream) {
    if (!out_params || !out_joints || !out_meta || !out_n_tracks || !next_id || !row_offsets || !message) return MVMC_ERR_ARG;
    if (n_frames < 0 || chain_len <= 0 || n_frames % chain_len || t_max <= 0 || t_max > ST_T || row_cap < 0) return MVMC_ERR_ARG;
    if (n_frames / chain_len > n_chains_cap) return MVMC_ERR_ARG;
    MsgLayout lay{n_chains_cap, t_max, row_cap};
    hipStream_t s = (hipStream_t)stream;
    if (n_frames == 0) {
        hipLaunchKernelGGL(empty_header_kernel, dim3(1), dim3(64), 0, s, chain_len, t_max, row_cap, (int32_t*)message);
        MVMC_CHECK_LAUNCH();
        return MVMC_OK;
    }
    // row_offsets: (n_frames + 1) words, the last one receives the total
    hipLaunchKernelGGL(row_offsets_kernel, dim3(1), dim3(1024), 0, s, out_n_tracks, n_frames, t_max, row_offsets, row_offsets + n_frames);
    hipLaunchKernelGGL(pack_kernel, dim3(n_frames), dim3(64), 0, s, out_params, out_joints, out_meta, out_n_tracks, next_id, row_offsets,
                       row_offsets + n_frames, n_frames, chain_len, t_max, lay, (uint32_t*)message);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}

extern "C" int mvmc_stitch_chains(const void* messages, long long message_words, int world, int n_chains_cap, int t_max, int row_cap,
                                  int id_cap, double max_dist, int n_chains_total_cap, int32_t* gid, int32_t* match, int32_t* info,
                                  int32_t* work, mvmcStream_t stream) {
    if (!messages || !gid || !match || !info || !work) return MVMC_ERR_ARG;
    if (world <= 0 || world > 64 || t_max <= 0 || t_max > ST_T || id_cap <= 0 || id_cap > 64 || n_chains_total_cap < 0) return MVMC_ERR_ARG;
    MsgLayout lay{n_chains_cap, t_max, row_cap};
    if (message_words < (long long)lay.words()) return MVMC_ERR_ARG;
    if ((long long)world * n_chains_cap > n_chains_total_cap) return MVMC_ERR_ARG;
    StitchArgs A;
    A.msgs = (const uint32_t*)messages; A.msg_words = (size_t)message_words; A.world = world; A.id_cap = id_cap; A.lay = lay;
    A.max_dist = max_dist; A.gid = gid; A.match = match; A.info = info;
    A.ptr = work; A.rank_of = work + (size_t)n_chains_total_cap * id_cap;
    if (hipMemsetAsync(info, 0, 4 * sizeof(int32_t), (hipStream_t)stream) != hipSuccess) return MVMC_ERR_LAUNCH;
    const int cap = world * n_chains_cap;
    if (cap > 0) hipLaunchKernelGGL(stitch_match_kernel, dim3((cap + 63) / 64), dim3(64), 0, (hipStream_t)stream, A);
    hipLaunchKernelGGL(stitch_ids_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, A);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}
