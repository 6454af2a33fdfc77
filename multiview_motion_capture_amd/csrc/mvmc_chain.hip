// The temporal hot path with the chains decoupled: ONE persistent 256-thread workgroup per chain runs
//   graph (match_spatial / match_spatial_time) -> ALS -> assignment -> IK (one wave per person) -> commit
// for all frames of its chain, with no launch boundaries in between (MvTracker.update_4d, motion_capture.py:873-963,
// per chain; tracker.ChainTracker.step is the launch-per-stage form of the same sequence).
//
// Why: in the launch-per-stage form every launch lasts as long as its slowest member -- the ALS launch as long as the
// graph with the most iterations (~290 against a mean of 140), the IK launch as long as the slowest solve -- and the
// chip idles meanwhile.  Chains are independent, so here each advances at its own pace: a chain's time is the sum of
// its OWN graph and solve times, and all chains are resident at once (3 workgroups per CU: 768 slots).
//
// The device code is the code of the separate kernels (affinity_wave, st_affinity_wave, als4_graph, assign_chain,
// ik1_solve, commit_chain), so the results are the same bit for bit; LDS is one arena reused by the phases:
//   [ graph scratch | Als4Lds | 4 x Ik1Shared ]  (union)  +  Ik1Tables (persistent)
#define MVMC_DEVICE_ONLY
#if !defined(MVMC_CHAIN_BIG_TU) && !defined(MVMC_SMALL_WPS)
#define MVMC_SMALL_WPS 4   // the SMALL layout: 128 VGPRs, four workgroups per CU (mvmc_common.h)
#endif
// Wave priority by PHASE (SMALL layout; s_setprio): graph, association, assignment, commit and outputs run at priority 1, the IK at 0.
// The association is a latency chain (one solver wave per workgroup on the critical path, the workers mostly waiting), the IK is
// issue-bound; with four workgroups per CU sharing the SIMDs an association iteration took 12.7 k cycles against 7.4 k alone.  With
// the priority its phase costs 18.4 M cycles per chain instead of 28.5 M, the IK 41.7 M instead of 33.9 M: a chain 65.0 -> 62.3 M,
// 513 k -> 530 k frames/s (same box).  Priority 3 instead of 1, or the solver wave above its workers: no further change.
#if !defined(MVMC_CHAIN_BIG_TU) && !defined(MVMC_PRIO_ALS) && !defined(MVMC_NO_PHASE_PRIO)
#define MVMC_PRIO_ALS 1
#define MVMC_PRIO_REST 1
#endif
// Two stage-1 IK models on one wave (mvmc_ik_pair.h), for the throughput build of the SMALL layout: built and measured in round 6, NOT
// shipped (-DMVMC_WITH_IK_PAIR builds it).  Bit-identical; 96 % of the stage-1 models of a warm solve are made in pairs, the launch
// issues 10 % fewer instructions -- and runs 1.5 - 2 % SLOWER (541 - 544 k against 552 - 553 k frames/s, same box): a wave's dependent chain,
// not the issue rate, bounds the IK phase, and a pair costs the later of its two waves nothing while the earlier one waits
// (docs/design_measurement.md, "Round 6").
#if defined(MVMC_WITH_IK_PAIR) && !defined(MVMC_CHAIN_BIG_TU) && !defined(MVMC_CHAIN_LAT_TU) && MVMC_SMALL_WPS >= 4
#define MVMC_IK_PAIR 1
#endif
#include <cstdlib>
#include "mvmc_common.h"
#include "mvmc_assoc.hip"
#include "mvmc_track.hip"
#include "mvmc_ik1.hip"

// kernel arguments (a named type with external linkage: the BIG layout's launcher lives in another translation unit)
struct MvmcChainArgs {
    // inputs
    const double* kps17;      // (F,C,P,17,3), F = n_chains * L, chain b owns frames [b L, (b+1) L)
    const int32_t* counts;    // (F,C)
    const double* Pm;         // (C,3,4)
    const float* Fm;          // (C,C,3,3) f32 (match_spatial)
    const double* F2;         // (C,C,3,3) f64 (match_spatial_time)
    const double* seed;       // RandomState(0).rand() table
    int seed_len;
    int n_chains, L, C, P, T, K, V, nfev_cold, nfev_warm, n_inits;
    // tracker state per chain
    double* params;           // (B,T,68)
    double* joints;           // (B,T,18,3)
    int32_t* meta;            // (B,T,4)
    int32_t* n_tracks;        // (B)
    int32_t* next_id;         // (B)
    int32_t* n_dead;          // (B)
    int32_t* slot_src;        // (B,T)
    // per-chain workspaces
    float* S_sp;              // (B,N,N)          N = C P
    double* W_st;             // (B,NS,NS)        NS = T + C P
    int32_t* gc;              // (B,C+1)
    int32_t* labels_sp;       // (B,N)
    int32_t* labels_st;       // (B,NS)
    int32_t* ncl_sp;          // (B)
    int32_t* ncl_st;          // (B)
    int32_t* iters_sp;        // (B)
    int32_t* iters_st;        // (B)
    int32_t* members;         // (B,NP,V)         NP = T + K; row s valid in [0, n_members[s])
    int32_t* n_members;       // (B,NP)
    uint8_t* cold;            // (B,NP)
    double* init;             // (B,NP,68)
    int32_t* status;          // (B,T)
    int32_t* n_new;           // (B)
    double* ik_params;        // (B,NP,68)
    double* ik_joints;        // (B,NP,54)
    double* ik_info;          // (B,NP,8)
    double* ik_scratch;       // (B,8,MVMC_IK_SCRATCH_DOUBLES): one block per IK wave (4 in the SMALL layout, 8 in the BIG one)
    // per-frame outputs
    double* out_params;       // (F,T,68)
    double* out_joints;       // (F,T,54)
    int32_t* out_meta;        // (F,T,4)
    int32_t* out_n;           // (F)
    double* out_info;         // (F,NP,8) or NULL
    int32_t* out_iters;       // (F) ALS iterations of the frame's graph, or NULL
    double* out_cycles;       // (B,8) shader cycles by phase {graph, ALS, assign, IK, commit, outputs, total}, or NULL
    int parts;                // workgroups per chain (consecutive frame ranges, handed over through flags)
    int queue;                // 0: workgroup (part, chain) = block index (part * n_chains + chain), a part waits for its chain's flag;
                              // 1: a workgroup draws a ticket when it starts and takes the chain that has been ready longest;
                              // 2: a workgroup draws a ticket when it starts and (part, chain) = ticket, as in 0
    unsigned* flags;          // (2 B + 4): [0,B) parts completed per chain; [B] time-out, [B+1] graph too large, [B+2] capacity word of
                              // the launch; [B+4+b] the void word of chain b (bit 0 views / clusters, bit 1 tracklet table, bit 2 graph
                              // too large for the layout's association variant); [2B+4] ticket counter, [2B+5] ring tail,
                              // [2B+6 ...) ready ring of B * (parts - 1) entries (queue mode).  Zeroed by the launcher
    int self_zero;            // 1: ONE chain run by ONE workgroup of the latency build zeroes the words itself (no memset in front of
                              // the launch: a device operation less per frame of MvTracker.update_4d)
};


namespace {

using ChainArgs = MvmcChainArgs;
// The launch's arguments as the device functions see them: a reference INTO THE KERNEL-ARGUMENT SEGMENT (constant address space), so
// every field is a scalar load where it is used.  Handing the phases a reference to the kernel's by-value parameter made the compiler
// copy the 368-byte struct to the stack of every lane at kernel entry (24 sixteen-byte scratch stores per lane: 98 KB per workgroup,
// i.e. per frame -- 1 GB per launch) and read every field back from scratch.
using ChainArgsK = const __attribute__((address_space(4))) MvmcChainArgs;

// Two layouts.  SMALL (configs 1-4: N = C P <= 40 nodes, <= 6 views per person): the rank-8 workgroup ALS variants, 52 KB of LDS,
// three workgroups per CU.  BIG (config 5, C8 P8: N <= 64, N + T <= 72, <= 8 views per person): the generic workgroup ALS (rank
// <= 16: als5_graph, X1 as a dense n x n matrix in LDS, the element state in registers) on a 512-thread workgroup -- eight waves, so the
// eight people of a frame are solved side by side -- with 112 KB of LDS in a dynamic allocation, one workgroup per CU.
template <bool BIG> struct ChainCfg;
// POOL = view blocks (pose + projection of one member of a cluster) the IK phase has room for, for ALL problems of a frame together: the
// clusters of a frame are disjoint, so a frame with at most POOL poses never runs out, whatever the size of a single cluster
// (SMALL: the association variants hold 24 pose nodes anyway; BIG: N_MAX).
template <> struct ChainCfg<false> { static constexpr int POOL = 24, N_MAX = 40, NS_MAX = 48, NT = 256, WG_PER_CU = MVMC_SMALL_WPS, WAVES_PER_SIMD = MVMC_SMALL_WPS; };
// (BIG, round 5: sixteen tracklet slots, graphs of up to 80 nodes.  The fast association variant -- als5: n <= 72, rank <= 16, i.e. eight
// live tracklets at C8 P8 -- runs wherever the frame's graph fits it; a frame with a ninth tracklet -- rank 2 x 9 = 18, n = 73 -- takes
// the generic variant (n <= 80, rank <= 32) IN the same workgroup, so its chain stays in the launch instead of going through the
// repair tier: that frame's association is ~3 x slower, the chain is not re-run stage by stage.)
template <> struct ChainCfg<true> { static constexpr int POOL = 64, N_MAX = 64, NS_MAX = 80, NT = 512, WG_PER_CU = 1, WAVES_PER_SIMD = 2; };


constexpr int CH_EOFF = 2368;   // SMALL: doubles of graph scratch in front of the pose-pair block (NS <= 48: 48 * 48 + 6 + 48 = 2358)
constexpr int CH_KOFF_BIG = 6496;                  // BIG: the frame's keypoints behind st_affinity_wave's part (80 * 80 + 10 + 80 = 6490)
constexpr int CH_EOFF_BIG = CH_KOFF_BIG + 64 * 51; // BIG: the pose-pair block (64 x 64) behind the keypoints

template <bool BIG> union ChainArena;
template <> union ChainArena<false> {
    Als4Lds<32> als_st;
    Als4Lds<24> als_sp;
    struct { Ik1Shared ik[4]; int mq[24]; unsigned short mc[24]; } ikp;   // four solves + the frame's member list (ChainCfg::POOL)
    // graph scratch: st_affinity_wave needs (NS*NS + 6) doubles + 2 NS ints (NS <= 48), affinity_wave N*51 doubles + 2 N*N floats
    // + 2 N ints + 4 words (N <= 40)
    // + the pose-pair block made ahead of the hand-over (st_pose_pairs: N * N doubles behind st_affinity_wave's part)
    double graph[CH_EOFF + 40 * 40];
};
template <> union ChainArena<true> {
    Als5Lds<72> als;
    AlsGenLds<80, 32, 512> als_wide;   // graphs beyond als5 (a ninth tracklet and more: rank 18 .. 32, n <= 80)
    struct { Ik1Shared ik[8]; int mq[64]; unsigned short mc[64]; } ikp;   // eight waves: the eight people of config 5 are solved side by side
    // affinity_wave at N = 64: 64 * 51 + 64 * 64 + 64 + 16 = 7440; st_affinity_wave at NS = 80: 80 * 80 + 6 + 80, + the frame's keypoints
    // + the pose-pair block; the line tables of st_pose_pairs_lines (2 * 4 * P * 68 doubles) use st_affinity_wave's part
    double graph[CH_EOFF_BIG + 64 * 64];
};
static_assert(CH_EOFF >= 48 * 48 + 10 + 48 && CH_EOFF + 40 * 40 >= 40 * 51 + 40 * 40 + 40 + 8, "graph scratch covers both graph builders");
static_assert(sizeof(ChainArena<false>) <= 4 * sizeof(Ik1Shared) + 144, "SMALL: the IK blocks set the arena size");
static_assert(sizeof(ChainArena<false>) + sizeof(Ik1Tables) + 8 <= 40960, "SMALL: four workgroups per CU (160 KB / 4, in granules of 1,280 B)");
static_assert(CH_EOFF_BIG + 64 * 64 >= 64 * 51 + 64 * 64 + 64 + 16 && CH_KOFF_BIG >= 80 * 80 + 10 + 80, "BIG: graph scratch covers both graph builders");
static_assert(sizeof(ChainArena<true>) <= 150 * 1024, "BIG: one workgroup per CU");

// The phases as separate (non-inlined) functions: each gets its own register allocation inside the workgroup's budget
// of 168 VGPRs (three workgroups per CU) instead of one allocation over the union of all phases.
// Every phase reports through `done`, a word in the KERNEL's stack frame: a call that is handed a pointer into its caller's frame is
// not a tail-call candidate, and only for such calls does the compiler drop the callee-saved register convention of these local
// functions -- with it each phase saved and restored, per call and per lane, every callee-saved vector register it touches (58 for the
// IK phase), whether the kernel had anything live there or not.
template <bool BIG>
__device__ __noinline__ void chain_graph_spatial(ChainArena<BIG>& arena_in, ChainArgsK& A, int b, int f, int* done) {
    ChainArena<BIG>& arena = *uni(&arena_in);
    MVMC_ASSUME_LDS(&arena);
    const int C = A.C, P = A.P, N = C * P;
    float* S = A.S_sp + (size_t)b * N * N;
    if ((threadIdx.x >> 6) == 0) affinity_wave(arena.graph, A.kps17, A.counts, A.Fm, C, P, f, nullptr, S);
    *done = 0;
}
template <bool BIG>
__device__ __noinline__ void chain_graph_temporal(ChainArena<BIG>& arena_in, ChainArgsK& A, int b, int f, bool pairs_ready, int* done) {
    ChainArena<BIG>& arena = *uni(&arena_in);
    MVMC_ASSUME_LDS(&arena);
    const int C = A.C, P = A.P, T = A.T, NS = T + C * P;
    double* W = A.W_st + (size_t)b * NS * NS;
    // BIG: the frame's keypoints through LDS (config 5: + 0.7 %).  SMALL keeps reading them from global memory: its pose-pair block is
    // made while the workgroup waits for its predecessor, where the staging pass is only more work (config 4: - 0.3 %)
    double* kf = nullptr;
    const double* Epre = (!BIG && pairs_ready) ? arena.graph + CH_EOFF : nullptr;
    if constexpr (BIG) {
        kf = arena.graph + CH_KOFF_BIG;
        st_stage_keypoints(kf, A.kps17, f, C, P);
        __syncthreads();
        if (2 * 4 * P * 68 <= CH_KOFF_BIG && C * P <= 64) {       // the pair errors by line tables (config 5: P = 8)
            st_pose_pairs_lines<ChainCfg<BIG>::NT>(arena.graph + CH_EOFF_BIG, arena.graph, kf, A.counts, f, A.F2, C, P, 0.1);
            Epre = arena.graph + CH_EOFF_BIG;
        }
    }
    st_affinity_wave<ChainCfg<BIG>::NT>(arena.graph, A.kps17, A.counts, 0, f, A.joints + (size_t)b * T * 54, A.n_tracks + b, A.Pm, A.F2, C, P,
                           T, 0.1, W, nullptr, A.gc + (size_t)b * (C + 1), Epre, C * P, kf);
    *done = 0;
}
// the frame's 2-D / 2-D distances, made while the workgroup waits for its predecessor (they do not depend on the tracklets)
__device__ __noinline__ void chain_pose_pairs(ChainArena<false>& arena_in, ChainArgsK& A, int f, int* done) {
    ChainArena<false>& arena = *uni(&arena_in);
    MVMC_ASSUME_LDS(&arena);
    st_pose_pairs(arena.graph + CH_EOFF, A.kps17, A.counts, f, A.F2, A.C, A.P, 0.1);
    *done = 0;
}
template <bool BIG>
__device__ __noinline__ void chain_als_spatial(ChainArena<BIG>& arena_in, ChainArgsK& A, int b, int f, int* done) {
    ChainArena<BIG>& arena = *uni(&arena_in);
    MVMC_ASSUME_LDS(&arena);
    const int C = A.C, N = C * A.P;
    // batch index 0 with pre-offset pointers: the graph, its group counts (the frame's people per view) and outputs
    if constexpr (BIG)
        als5_graph<float, 72>(arena.als, 0, A.S_sp + (size_t)b * N * N, A.counts + (size_t)f * C, C, N, A.seed, A.seed_len,
                                          nullptr, nullptr, A.labels_sp + (size_t)b * N, A.ncl_sp + b, A.iters_sp + b);
    else
        als4_graph<float, 24>(arena.als_sp, 0, A.S_sp + (size_t)b * N * N, A.counts + (size_t)f * C, C, N, A.seed, A.seed_len, nullptr,
                              nullptr, A.labels_sp + (size_t)b * N, A.ncl_sp + b, A.iters_sp + b);
    *done = 0;
}
template <bool BIG>
__device__ __noinline__ void chain_als_temporal(ChainArena<BIG>& arena_in, ChainArgsK& A, int b, int* done) {
    ChainArena<BIG>& arena = *uni(&arena_in);
    MVMC_ASSUME_LDS(&arena);
    const int C = A.C, NS = A.T + C * A.P;
    if constexpr (BIG) {
        // the frame's graph: n nodes, rank 2 x its largest group (mv_association.py:251-269); wave-uniform (every lane reads the same words)
        const int32_t* gc = A.gc + (size_t)b * (C + 1);
        int n = 0, gmax = 0;
        for (int g = 0; g <= C; ++g) { int c = gc[g]; c = c < 0 ? 0 : c; n += c; gmax = c > gmax ? c : gmax; }
        n = uni(n); gmax = uni(gmax);
        const int r = 2 * gmax < n ? 2 * gmax : n;
        if (n <= 72 && r <= 16)
            als5_graph<double, 72>(arena.als, 0, A.W_st + (size_t)b * NS * NS, gc, C + 1, NS, A.seed,
                                               A.seed_len, nullptr, nullptr, A.labels_st + (size_t)b * NS, A.ncl_st + b, A.iters_st + b);
        else
            als_gen_graph<double, 80, 32, 512>(arena.als_wide, 0, A.W_st + (size_t)b * NS * NS, gc, C + 1, NS, A.seed,
                                               A.seed_len, nullptr, nullptr, A.labels_st + (size_t)b * NS, A.ncl_st + b, A.iters_st + b);
    } else
        als4_graph<double, 32>(arena.als_st, 0, A.W_st + (size_t)b * NS * NS, A.gc + (size_t)b * (C + 1), C + 1, NS, A.seed, A.seed_len,
                               nullptr, nullptr, A.labels_st + (size_t)b * NS, A.ncl_st + b, A.iters_st + b);
    *done = 0;
}
__device__ __noinline__ void chain_assign(ChainArgsK& A, int b, int f, int* done) {
    assign_chain(threadIdx.x & 63, 64, b, f, A.labels_sp, A.ncl_sp, A.labels_st, A.ncl_st, A.counts, A.n_tracks, A.params, A.C, A.P, A.T, A.K, A.V,
                 A.members, A.cold, A.init, A.status, A.n_new, reinterpret_cast<int32_t*>(A.flags + A.n_chains + 4 + b), A.n_members);
    *done = 0;
}
__device__ __noinline__ void chain_commit(ChainArgsK& A, int b, int* done) {
    commit_chain(threadIdx.x & 63, 64, b, A.status, A.n_new, A.ik_params, A.ik_joints, A.T, A.K, A.n_inits, A.params, A.joints, A.meta, A.n_tracks,
                 A.next_id, A.n_dead, A.slot_src, reinterpret_cast<int32_t*>(A.flags + A.n_chains + 4 + b));
    *done = 0;
}
template <bool BIG>
__device__ __noinline__ void chain_ik(ChainArena<BIG>& arena_in, const Ik1Tables& tables, ChainArgsK& A, int b, int* done) {
    ChainArena<BIG>& arena = *uni(&arena_in);
    MVMC_ASSUME_LDS(&arena);
    MVMC_ASSUME_LDS(&tables);
    constexpr int NW = ChainCfg<BIG>::NT / 64, POOL = ChainCfg<BIG>::POOL;
    const int lane = threadIdx.x & 63;
    const int wave = uni((int)(threadIdx.x >> 6)), NP = A.T + A.K;
    // the view blocks of the frame's problems lie end to end in the pool: block of slot s starts at the members of the slots before it
    const int cnt = lane < NP ? A.n_members[(size_t)b * NP + lane] : 0;
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
    const int excl = incl - cnt;
    // Two stage-1 models on one wave (mvmc_ik_pair.h): the waves (0, 1), (2, 3) are pairs.  The pair words lie in the solve blocks, which
    // the other phases overwrite: mailbox empty, both waves away, before any solve of the frame starts.
    int pair_dS = 0, pair_dhh = 0;
#ifdef MVMC_IK_PAIR
    if constexpr (!BIG) {
        pair_dS = (wave & 1) ? -(int)sizeof(Ik1Shared) : (int)sizeof(Ik1Shared);
        pair_dhh = (wave & 1) ? -MVMC_IK_SCRATCH_DOUBLES : MVMC_IK_SCRATCH_DOUBLES;
        if (lane == 0) arena.ikp.ik[wave].pairw = (wave & 1) ? (3 << 16) : 0;
        __syncthreads();
    }
#endif
    // wave w takes the problem slots w, w + NW, ... of this chain
    for (int s = wave; s < NP; s += NW) {
        const int p = b * NP + s;
        const int base = uni(__shfl(excl, s, 64)), n_valid = uni(__shfl(cnt, s, 64));
        const int room = base < POOL ? POOL - base : 0;     // (a frame with more poses than POOL: the problem is cut short and flagged)
        ik1_solve(arena.ikp.ik[wave], arena.ikp.mq + (base < POOL ? base : 0), arena.ikp.mc + (base < POOL ? base : 0), room, tables, A.kps17, A.Pm, A.members,
                  p, A.V, A.C, A.P, A.init, A.cold, A.nfev_cold, A.nfev_warm, A.ik_params, A.ik_joints, A.ik_info,
                  A.ik_scratch + (ptrdiff_t)(b * NW + wave - p) * MVMC_IK_SCRATCH_DOUBLES, 3, nullptr,
                  reinterpret_cast<int32_t*>(A.flags + A.n_chains + 4 + b), n_valid, pair_dS, pair_dhh, wave & 1);
    }
    *done = 0;
}

template <bool BIG>
__global__ void __launch_bounds__(ChainCfg<BIG>::NT, ChainCfg<BIG>::WAVES_PER_SIMD)
chain_kernel(Ik1Tables tables_arg, ChainArgs A_by_value) {
    // A = the second kernel argument where it lies in the kernel-argument segment (layout: the tables, then A at its natural alignment)
    constexpr size_t A_OFFSET = (sizeof(Ik1Tables) + alignof(ChainArgs) - 1) / alignof(ChainArgs) * alignof(ChainArgs);
    typedef const __attribute__((address_space(4))) char* KernargBytes;
    ChainArgsK& A = *(ChainArgsK*)((KernargBytes)__builtin_amdgcn_kernarg_segment_ptr() + A_OFFSET);
    extern __shared__ __attribute__((aligned(16))) unsigned char chain_lds[];   // the arena (BIG: 72 KB, beyond the static limit)
    // The arena's address is made OPAQUE here: every call site hands the phases the same pointer, so interprocedural constant propagation
    // replaces their parameter by the dynamic-LDS symbol itself -- and an out-of-line function finds a dynamic allocation's offset by a
    // look-up in a table in GLOBAL memory (llvm.amdgcn.dynlds.offset.table), re-done wherever the allocator would rather reload than
    // keep a register: two global loads per iteration inside the association's loop, in front of its LDS reads.  As an opaque value it
    // is an ordinary argument, which the phases pin to scalar registers (uni).
    ChainArena<BIG>* arena_ptr = reinterpret_cast<ChainArena<BIG>*>(chain_lds);
    asm volatile("" : "+s"(arena_ptr));
    ChainArena<BIG>& arena = *arena_ptr;
    __shared__ Ik1Tables tables;
    __shared__ int s_nt;
    // Which (chain, part) a workgroup runs -- two protocols, the results are the same bit for bit:
    //  * static (A.queue == 0): block index = part * n_chains + chain.  Every workgroup of part p is dispatched before any of part
    //    p + 1 IF dispatch follows the block index, so a waiting workgroup's predecessor is resident or finished; later parts start
    //    wherever a slot frees up, and the CUs that run slower simply receive fewer of them.  The wait is bounded (4 s, loud), but the
    //    order of dispatch is not something the programming model promises.
    //  * queue (A.queue == 1): the workgroup draws a ticket when it STARTS; the first n_chains tickets are part 0 of the chains, ticket
    //    h takes entry h - n_chains of a ring that the workgroups finishing a non-final part fill in the order they finish (entry =
    //    chain and next part).  Nothing depends on the dispatch order.  No deadlock: a waiting ticket h needs h - n_chains + 1
    //    finished non-final parts; all lower tickets have started (a ticket is drawn by a RUNNING workgroup), and while a chain is
    //    unfinished they cannot all be final parts.  It costs 0.7 % on one GPU (the pose-pair block below can no longer be made during
    //    the wait).
    //  * ticket (A.queue == 2, the default): the static mapping, indexed by a TICKET drawn when the workgroup starts instead of by the
    //    block index: part = ticket / n_chains, chain = ticket % n_chains.  The predecessor of ticket h is ticket h - n_chains: a lower
    //    ticket, i.e. drawn by a workgroup that has already started -- running or finished, never waiting to be dispatched -- whatever
    //    order the dispatcher follows; by induction the lowest waiting ticket always has a running predecessor chain, so nothing can
    //    deadlock.  Chain and part are known at once, so the pose-pair block is made during the wait as in the static mapping, and with
    //    in-order dispatch the two mappings coincide: the static mapping's speed without its assumption, for one atomic per workgroup.
    const int tid = threadIdx.x, wave = tid >> 6;
#ifdef MVMC_CHAIN_LAT_TU
    if (A.self_zero) {       // (mvmc_chain_run: one chain, one workgroup -- the launcher sent no memset)
        if (tid < 2 * A.n_chains + 8) A.flags[tid] = 0u;
        __syncthreads();
    }
#endif
#ifdef MVMC_CHAIN_WAITPROF   // diagnostic build: out_cycles[7] = cycles a chain's workgroups were resident before their frames began
    const long long t_entry = clock64();
#endif
    __shared__ int s_task;   // part << 20 | chain, or -1: give up
    unsigned* const qwords = A.flags + 2 * A.n_chains + 4;   // {ticket, tail, ring[]}
    const bool queue = A.queue == 1 && A.parts > 1;
    const bool by_ticket = A.queue == 2 && A.parts > 1;
    unsigned ticket = 0;
    if ((queue || by_ticket) && tid == 0) ticket = __hip_atomic_fetch_add(qwords, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (by_ticket) {                       // every thread needs the index: through s_nt (next written inside the frame loop)
        if (tid == 0) s_nt = (int)ticket;
        __syncthreads();
    }
    // The skeleton tables (ancestor masks, level lists, active columns, row masks) are made ONCE PER CALL on the host by the launcher
    // (ik1_build_tables_host: the same source as the device's ik1_build_tables) and arrive as the first kernel argument: a word per
    // lane from the kernel-argument segment into LDS.  Built here by one wave of every workgroup they cost 97 k cycles per frame --
    // hidden while a workgroup waits for its predecessor, but 2.4 % of the resident time when two launches share the GPU and the waits
    // are short (tools/chain_wait_probe.py): 431.7 k -> 437.1 k frames/s on the same box, bit-identical.
    {
        static_assert(sizeof(Ik1Tables) % 4 == 0, "copied by words");
        typedef const __attribute__((address_space(4))) unsigned* KernargWords;
        KernargWords src = (KernargWords)__builtin_amdgcn_kernarg_segment_ptr();
        unsigned* dst = reinterpret_cast<unsigned*>(&tables);
        for (int i = tid; i < (int)(sizeof(Ik1Tables) / 4); i += ChainCfg<BIG>::NT) dst[i] = src[i];
    }
    int b, part;
    int done = 0;   // the phases' report word (see above)
    if (!queue) {
        const int idx = by_ticket ? uni(s_nt) : (int)blockIdx.x;
        b = idx % A.n_chains; part = idx / A.n_chains;
        // work that does not depend on the chain's state comes before the hand-over: for a workgroup that has a predecessor, the
        // pose-pair block of its first frame's graph
        if constexpr (!BIG) { if (part > 0) chain_pose_pairs(arena, A, b * A.L + part * A.L / A.parts, &done); }
    }
    if (queue || part > 0) {
        // consumer side of the hand-off (cdna_hip_programming.md Guideline 16): one lane polls ONE word relaxed (the chain's flag, or
        // its ticket's ring entry), one agent-scope acquire, then the workgroup's barrier; the chain state is read with plain vector
        // loads after it
        if (tid == 0) {
            int task = queue ? (int)ticket : ((part << 20) | b);       // queue: ticket < n_chains = part 0 of chain `ticket`
            const bool must_wait = queue ? ticket >= (unsigned)A.n_chains : true;
            if (must_wait) {
                const unsigned* word = queue ? qwords + 2 + (ticket - (unsigned)A.n_chains) : A.flags + b;
                const unsigned need = queue ? 1u : (unsigned)part;      // ring entries are task + 1 (0 = not yet filled)
                // bounded by wall time (~4 s at the 100 MHz constant clock), not by a spin count; a time-out anywhere in the launch
                // (the error word) also ends this wait, so a chain of waiting parts does not pay the time-out once per part
                const unsigned long long t0 = wall_clock64();
                bool abort = false;
                unsigned spins = 0, v;
                while ((v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need) {
                    __builtin_amdgcn_s_sleep(32);
                    // the launch-wide error word is ONE address for every waiting workgroup of the launch: looked at once in 1024 polls
                    // (hundreds of pollers on one line cost the whole chip memory bandwidth: measured 370 k -> 331 k frames/s)
                    if ((++spins & 1023u) != 0u) continue;
                    if (__hip_atomic_load(A.flags + A.n_chains, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { abort = true; break; }
                    if (wall_clock64() - t0 > 400000000ull) {   // static: dispatch did not come in block order; queue: nothing became ready
                        __hip_atomic_store(A.flags + A.n_chains, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        abort = true;
                        break;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (queue && !abort) task = (int)v - 1;
                if (abort) {
                    // release the successors: in the static protocol they must not wait for this part's flag (in the queue protocol every
                    // waiter sees the error word within 1024 polls)
                    if (!queue) __hip_atomic_store(A.flags + b, (unsigned)(part + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    task = -1;
                }
            }
            s_task = task;     // ONE lane decides; the whole workgroup branches on the same value
        }
        __syncthreads();
        const int task = uni(s_task);
        if (task < 0) return;
        b = task & 0xFFFFF; part = task >> 20;
        if constexpr (!BIG) { if (queue && part > 0) chain_pose_pairs(arena, A, b * A.L + part * A.L / A.parts, &done); }
    }
    const int T = A.T, NP = T + A.K;
    const int t_lo = part * A.L / A.parts, t_hi = (part + 1) * A.L / A.parts;
    __syncthreads();
    long long cyc[6] = {0, 0, 0, 0, 0, 0}, t_prev = clock64();
    const long long t_start = t_prev;
    auto lap = [&](int k) { const long long now = clock64(); cyc[k] += now - t_prev; t_prev = now; };
    for (int t = t_lo; t < t_hi; ++t) {
        const int f = b * A.L + t;
        if (tid == 0) s_nt = mvmc_ld_i32(A.n_tracks + b);
        __syncthreads();
        const int nt = uni(s_nt);
        // ---- graph + association ----
#ifdef MVMC_PRIO_ALS
        if constexpr (!BIG) __builtin_amdgcn_s_setprio(MVMC_PRIO_ALS);
#endif
        if (nt <= 0) {   // no live tracklets: match_spatial (motion_capture.py:597-631), f32 affinity
            chain_graph_spatial<BIG>(arena, A, b, f, &done);
            __syncthreads();
            lap(0);
            chain_als_spatial<BIG>(arena, A, b, f, &done);
        } else {
            chain_graph_temporal<BIG>(arena, A, b, f, part > 0 && t == t_lo, &done);   // (the pose-pair block of a part's first frame is ready)
            __syncthreads();
            lap(0);
            chain_als_temporal<BIG>(arena, A, b, &done);
        }
#if defined(MVMC_PRIO_ALS) && !defined(MVMC_PRIO_REST)
        if constexpr (!BIG) __builtin_amdgcn_s_setprio(0);
#endif
        __syncthreads();
        lap(1);
        // a graph with more nodes (or a higher rank) than the workgroup variant of the ALS holds is flagged by it (iters < 0):
        // raise the launch's error word instead of silently tracking nobody
        if (tid == 0 && (nt <= 0 ? A.iters_sp[b] : A.iters_st[b]) < 0) {
            __hip_atomic_store(A.flags + A.n_chains + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicOr(A.flags + A.n_chains + 4 + b, 4u);
        }
        if (wave == 0) chain_assign(A, b, f, &done);   // clusters -> IK problems (bulk copies on the wave, the logic on lane 0)
        __syncthreads();
        lap(2);
#ifdef MVMC_PRIO_REST
        if constexpr (!BIG) __builtin_amdgcn_s_setprio(0);
#endif
        chain_ik<BIG>(arena, tables, A, b, &done);
#ifdef MVMC_PRIO_REST
        if constexpr (!BIG) __builtin_amdgcn_s_setprio(MVMC_PRIO_REST);
#endif
        __syncthreads();
        lap(3);
        if (wave == 0) chain_commit(A, b, &done);      // tracklet table after the frame
        __syncthreads();
        lap(4);
        // ---- per-frame outputs ----
        for (int e = tid; e < T * 68; e += ChainCfg<BIG>::NT) A.out_params[(size_t)f * T * 68 + e] = A.params[(size_t)b * T * 68 + e];
        for (int e = tid; e < T * 54; e += ChainCfg<BIG>::NT) A.out_joints[(size_t)f * T * 54 + e] = A.joints[(size_t)b * T * 54 + e];
        for (int e = tid; e < T * 4; e += ChainCfg<BIG>::NT) A.out_meta[(size_t)f * T * 4 + e] = A.meta[(size_t)b * T * 4 + e];
        if (A.out_info)
            for (int e = tid; e < NP * 8; e += ChainCfg<BIG>::NT) A.out_info[(size_t)f * NP * 8 + e] = A.ik_info[(size_t)b * NP * 8 + e];
        if (tid == 0) {
            A.out_n[f] = mvmc_ld_i32(A.n_tracks + b);
            if (A.out_iters) A.out_iters[f] = nt <= 0 ? A.iters_sp[b] : A.iters_st[b];
        }
        __syncthreads();
        lap(5);
    }
    // the chain's void word (assignment, IK pool, commit) into the launch's capacity word
    if (tid == 0) {
        const unsigned v = __hip_atomic_load(A.flags + A.n_chains + 4 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xBu;   // (bit 3: mvmc_ik_pair.h's net)
        if (v) atomicOr(A.flags + A.n_chains + 2, v);
    }
    if (done != 0) return;   // (never: the phases write 0)
    if (A.out_cycles && tid == 0) {   // accumulated over the chain's parts (they run one after the other)
        double* oc = A.out_cycles + (size_t)b * 8;
        for (int k = 0; k < 6; ++k) oc[k] = (part ? oc[k] : 0.0) + (double)cyc[k];
        oc[6] = (part ? oc[6] : 0.0) + (double)(clock64() - t_start);
#ifdef MVMC_CHAIN_WAITPROF
        oc[7] = (part ? oc[7] : 0.0) + (double)(t_start - t_entry);
#else
        oc[7] = (double)(part + 1);
#endif
    }
    if (A.parts > 1) {
        // producer side: every wave drains its stores, the barrier, then one lane releases at agent scope and raises the flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(A.flags + b, (unsigned)(part + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (queue && part + 1 < A.parts) {   // the chain's next part may run: the next free ring entry (one per finished non-final part)
                const unsigned at = __hip_atomic_fetch_add(qwords + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(qwords + 2 + at, (unsigned)(((part + 1) << 20) | b) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

}  // namespace

// The two layouts are compiled in SEPARATE translation units (mvmc_chain_big.hip includes this file with MVMC_CHAIN_BIG_TU): the
// out-of-line device functions both kernels call (proj_dist, the pairwise sums, assign / commit, the IK model ...) take their register
// budget from the loosest kernel that reaches them, so in one unit the BIG kernel (one workgroup per CU, 512 VGPRs) let them grow to
// 248 VGPRs and dropped the SMALL kernel from three workgroups per CU to one (measured: 370 k -> 167 k frames/s on config 4).
int mvmc_chain_launch_big(const void* tables_host, const MvmcChainArgs& A, int n_blocks, hipStream_t stream);   // (Ik1Tables: a type of each unit)

// the SMALL layout built for 256 VGPRs (two workgroups per CU, mvmc_chain_lat.hip): what a launch of few workgroups runs -- a frame at
// a time through MvTracker.update_4d, short sequences --, where one chain's latency counts and the other workgroup slots stay empty anyway
int mvmc_chain_launch_small_lat(const void* tables_host, const MvmcChainArgs& A, int n_blocks, hipStream_t stream);

#if defined(MVMC_CHAIN_LAT_TU)
int mvmc_chain_launch_small_lat(const void* tables_host, const MvmcChainArgs& A, int n_blocks, hipStream_t stream) {
    static_assert(MVMC_SMALL_WPS == 2, "the latency build: 256 VGPRs, two workgroups per CU");
    hipLaunchKernelGGL(chain_kernel<false>, dim3(n_blocks), dim3(256), sizeof(ChainArena<false>), stream, *static_cast<const Ik1Tables*>(tables_host), A);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}
#elif defined(MVMC_CHAIN_BIG_TU)
int mvmc_chain_launch_big(const void* tables_host, const MvmcChainArgs& A, int n_blocks, hipStream_t stream) {
    // per call: the attribute belongs to the function ON THE CURRENT DEVICE, and a cached flag would be neither per device nor
    // thread-safe (it costs ~1 us beside a launch of hundreds of milliseconds)
    if (hipFuncSetAttribute((const void*)chain_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)sizeof(ChainArena<true>)) != hipSuccess)
        return MVMC_ERR_LAUNCH;
    hipLaunchKernelGGL(chain_kernel<true>, dim3(n_blocks), dim3(ChainCfg<true>::NT), sizeof(ChainArena<true>), stream, *static_cast<const Ik1Tables*>(tables_host), A);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}
#else
extern "C" int mvmc_chain_run(const mvmcSkeleton* skel_host, const mvmcChainBuffers* buf, mvmcStream_t stream) {
    if (!skel_host || !buf) return MVMC_ERR_ARG;
    const mvmcChainBuffers& B = *buf;
    if (B.n_chains < 0 || B.chain_len <= 0 || B.n_views <= 0 || B.p_max <= 0 || B.t_max <= 0 || B.k_max <= 0 || B.v_max <= 0)
        return MVMC_ERR_ARG;
    if (B.max_nfev_cold < 1 || B.max_nfev_warm < 1) return MVMC_ERR_ARG;
    // sizes the two LDS layouts are built for (the launch-per-stage path covers everything else).  SMALL: padded sizes N <= 40,
    // N + T <= 48, and a frame's ACTUAL graph must have <= 24 nodes on the match_spatial path and <= 32 on the match_spatial_time
    // path -- checked on the device, flags[n_chains + 1].  BIG (C8 P8): N <= 64, T <= 16, N + T <= 80: every graph of those sizes fits.
    const int N = B.n_views * B.p_max, NS = B.t_max + N;
    if (B.n_views > 16 || 2 * B.p_max > 16) return MVMC_ERR_UNSUPPORTED;
    if (B.t_max + B.k_max > 64 || B.v_max > 64) return MVMC_ERR_UNSUPPORTED;   // a lane per problem slot / per member
    // SMALL: rank 2 max(t_max, p_max) <= 16 (its association variants); BIG: up to sixteen tracklet slots (rank <= 32 in its wide variant)
    const bool small = N <= ChainCfg<false>::N_MAX && NS <= ChainCfg<false>::NS_MAX && !B.force_big &&
                       2 * (B.t_max > B.p_max ? B.t_max : B.p_max) <= 16;
    const bool big = N <= ChainCfg<true>::N_MAX && NS <= ChainCfg<true>::NS_MAX && B.t_max <= 16;
    if (!small && !big) return MVMC_ERR_UNSUPPORTED;
    const void* need[] = {B.kps17, B.counts, B.Pmats, B.Fmats, B.F2, B.seed_table, B.params, B.joints, B.meta, B.n_tracks,
                          B.next_id, B.n_dead, B.slot_src, B.S_sp, B.W_st, B.group_counts, B.labels_sp, B.labels_st, B.n_clusters_sp,
                          B.n_clusters_st, B.iters_sp, B.iters_st, B.members, B.n_members, B.cold, B.init, B.status, B.n_new, B.ik_params, B.ik_joints, B.ik_info, B.ik_scratch,
                          B.out_params, B.out_joints, B.out_meta, B.out_n_tracks, B.flags};
    for (const void* q : need)
        if (!q) return MVMC_ERR_ARG;
    if (B.n_chains == 0) return MVMC_OK;
    SkelDev sk;
    if (!skel_to_dev(skel_host, &sk)) return MVMC_ERR_ARG;
    if (sk.n_side != MVMC_N_SIDE) return MVMC_ERR_UNSUPPORTED;
    ChainArgs A;
    A.kps17 = B.kps17; A.counts = B.counts; A.Pm = B.Pmats; A.Fm = B.Fmats; A.F2 = B.F2; A.seed = B.seed_table;
    A.seed_len = B.seed_len;
    A.n_chains = B.n_chains; A.L = B.chain_len; A.C = B.n_views; A.P = B.p_max; A.T = B.t_max; A.K = B.k_max; A.V = B.v_max;
    A.nfev_cold = B.max_nfev_cold; A.nfev_warm = B.max_nfev_warm; A.n_inits = B.n_inits;
    A.params = B.params; A.joints = B.joints; A.meta = B.meta; A.n_tracks = B.n_tracks; A.next_id = B.next_id;
    A.n_dead = B.n_dead; A.slot_src = B.slot_src;
    A.S_sp = B.S_sp; A.W_st = B.W_st; A.gc = B.group_counts; A.labels_sp = B.labels_sp; A.labels_st = B.labels_st;
    A.ncl_sp = B.n_clusters_sp; A.ncl_st = B.n_clusters_st; A.iters_sp = B.iters_sp; A.iters_st = B.iters_st;
    A.members = B.members; A.n_members = B.n_members; A.cold = B.cold; A.init = B.init; A.status = B.status;
    A.n_new = B.n_new; A.ik_params = B.ik_params; A.ik_joints = B.ik_joints; A.ik_info = B.ik_info; A.ik_scratch = B.ik_scratch;
    A.out_params = B.out_params; A.out_joints = B.out_joints; A.out_meta = B.out_meta; A.out_n = B.out_n_tracks;
    A.out_info = B.out_info; A.out_iters = B.out_als_iters; A.out_cycles = B.out_phase_cycles;
    A.parts = B.n_parts > 1 ? B.n_parts : 1;
    if (A.parts > 1 && B.chain_len % A.parts != 0) return MVMC_ERR_ARG;
    if (B.hand_over < 0 || B.hand_over > 2) return MVMC_ERR_ARG;
    A.queue = B.hand_over;
    if (B.n_chains >= (1 << 20) || A.parts >= (1 << 10)) return MVMC_ERR_UNSUPPORTED;   // (a ring entry is part << 20 | chain)
    A.flags = B.flags;
    A.self_zero = 0;
    // Few workgroups (a frame at a time, short sequences): the 256-register build of the same kernel (mvmc_chain_lat.hip), see below
    bool lat = false;
    if (small) {
        static const bool no_lat = getenv("MVMC_CHAIN_NO_LAT") && atoi(getenv("MVMC_CHAIN_NO_LAT")) != 0;
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        }
        lat = !no_lat && (long long)B.n_chains * A.parts <= 2LL * cus;
    }
    if (lat && B.n_chains == 1 && A.parts == 1)
        A.self_zero = 1;     // (one workgroup: it zeroes its ten words itself, the per-frame driver saves a device operation)
    else if (hipMemsetAsync(B.flags, 0, sizeof(unsigned) * ((size_t)B.n_chains * (A.parts + 1) + 8), (hipStream_t)stream) != hipSuccess)
        return MVMC_ERR_LAUNCH;
    Ik1Tables tables_host;   // the skeleton's tables: once per call, on the host, a kernel argument of the launch
    ik1_build_tables_host(tables_host, sk);
    if (!small)
        return mvmc_chain_launch_big(&tables_host, A, B.n_chains * A.parts, (hipStream_t)stream);
    {
        // Few workgroups (a frame at a time, short sequences): the 256-register build of the same kernel (mvmc_chain_lat.hip) -- a
        // frame's dependent chain is ~10 % shorter with the larger batches and fewer spills of that build, and with at most two
        // workgroups per CU to place, the other slots buy nothing.  Same results bit for bit.  MVMC_CHAIN_NO_LAT=1: the 128-register build.
        if (lat)
            return mvmc_chain_launch_small_lat(&tables_host, A, B.n_chains * A.parts, (hipStream_t)stream);
    }
    // (MVMC_CHAIN_EXTRA_LDS: an occupancy experiment -- bytes of LDS nobody uses, so that fewer workgroups share a CU)
    static const size_t extra_lds = getenv("MVMC_CHAIN_EXTRA_LDS") ? (size_t)atoi(getenv("MVMC_CHAIN_EXTRA_LDS")) : 0;
    if (extra_lds && hipFuncSetAttribute((const void*)chain_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(sizeof(ChainArena<false>) + extra_lds)) != hipSuccess) return MVMC_ERR_LAUNCH;
    hipLaunchKernelGGL(chain_kernel<false>, dim3(B.n_chains * A.parts), dim3(256), sizeof(ChainArena<false>) + extra_lds, (hipStream_t)stream, tables_host, A);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}
#endif  // MVMC_CHAIN_BIG_TU
