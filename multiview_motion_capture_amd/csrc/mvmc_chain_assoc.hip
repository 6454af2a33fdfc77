// Kernel A of the split BIG layout (mvmc_chain_split.h): graph -> association -> assignment of a chain-frame, persistent, one workgroup
// per CU.  Its own translation unit: the out-of-line device functions take their register budget from the kernel of their unit
// (168 VGPRs here: two waves per SIMD beside one wave of kernel B).
#define MVMC_CHAIN_BIG_TU
#define MVMC_CHAIN_SPLIT_TU
#include "mvmc_chain.hip"
#include "mvmc_chain_split.h"

namespace {

union AssocArena {
    Als5Lds<72, false> als;             // (the symmetrised affinity in MvmcChainArgs::wsym)
    AlsGenLds<80, 20, 512> als_wide;    // graphs beyond als5: a ninth or tenth tracklet (rank 18 / 20, n <= 80)
    double graph[CH_EOFF_BIG + 64 * 64];
};
static_assert(sizeof(AssocArena) <= 114 * 1024, "kernel A's arena beside kernel B's 42 KB on one CU (160 KB)");

__device__ __noinline__ void assoc_graph_spatial(AssocArena& ar_in, ChainArgsK& A, int b, int f, int* done) {
    AssocArena& ar = *uni(&ar_in);
    MVMC_ASSUME_LDS(&ar);
    const int C = A.C, P = A.P, N = C * P;
    if ((threadIdx.x >> 6) == 0) affinity_wave(ar.graph, A.kps17, A.counts, A.Fm, C, P, f, nullptr, A.S_sp + (size_t)b * N * N);
    *done = 0;
}
__device__ __noinline__ void assoc_graph_temporal(AssocArena& ar_in, ChainArgsK& A, int b, int f, int* done) {
    AssocArena& ar = *uni(&ar_in);
    MVMC_ASSUME_LDS(&ar);
    const int C = A.C, P = A.P, T = A.T, NS = T + C * P;
    double* kf = ar.graph + CH_KOFF_BIG;
    const double* Epre = nullptr;
    st_stage_keypoints(kf, A.kps17, f, C, P);
    __syncthreads();
    if (2 * 4 * P * 68 <= CH_KOFF_BIG && C * P <= 64) {
        st_pose_pairs_lines<512>(ar.graph + CH_EOFF_BIG, ar.graph, kf, A.counts, f, A.F2, C, P, 0.1);
        Epre = ar.graph + CH_EOFF_BIG;
    }
    st_affinity_wave<512>(ar.graph, A.kps17, A.counts, 0, f, A.joints + (size_t)b * T * 54, A.n_tracks + b, A.Pm, A.F2, C, P, T, 0.1,
                          A.W_st + (size_t)b * NS * NS, nullptr, A.gc + (size_t)b * (C + 1), Epre, C * P, kf);
    *done = 0;
}
__device__ __noinline__ void assoc_als_spatial(AssocArena& ar_in, ChainArgsK& A, int b, int f, int* done) {
    AssocArena& ar = *uni(&ar_in);
    MVMC_ASSUME_LDS(&ar);
    const int C = A.C, N = C * A.P;
    als5_graph<float, 72, false>(ar.als, 0, A.S_sp + (size_t)b * N * N, A.counts + (size_t)f * C, C, N, A.seed, A.seed_len, nullptr, nullptr,
                                 A.labels_sp + (size_t)b * N, A.ncl_sp + b, A.iters_sp + b, A.wsym + (size_t)b * MVMC_WSYM_DOUBLES);
    *done = 0;
}
__device__ __noinline__ void assoc_als_temporal(AssocArena& ar_in, ChainArgsK& A, int b, int* done) {
    AssocArena& ar = *uni(&ar_in);
    MVMC_ASSUME_LDS(&ar);
    const int C = A.C, NS = A.T + C * A.P;
    const int32_t* gc = A.gc + (size_t)b * (C + 1);
    int n = 0, gmax = 0;
    for (int g = 0; g <= C; ++g) { int c = gc[g]; c = c < 0 ? 0 : c; n += c; gmax = c > gmax ? c : gmax; }
    n = uni(n); gmax = uni(gmax);
    const int r = 2 * gmax < n ? 2 * gmax : n;
    if (n <= 72 && r <= 16)
        als5_graph<double, 72, false>(ar.als, 0, A.W_st + (size_t)b * NS * NS, gc, C + 1, NS, A.seed, A.seed_len, nullptr, nullptr,
                                      A.labels_st + (size_t)b * NS, A.ncl_st + b, A.iters_st + b, A.wsym + (size_t)b * MVMC_WSYM_DOUBLES);
    else
        als_gen_graph<double, 80, 20, 512>(ar.als_wide, 0, A.W_st + (size_t)b * NS * NS, gc, C + 1, NS, A.seed, A.seed_len, nullptr, nullptr,
                                           A.labels_st + (size_t)b * NS, A.ncl_st + b, A.iters_st + b);
    *done = 0;
}

__global__ void __launch_bounds__(512, 3)
chain_assoc_kernel(ChainArgs A_by_value) {
    ChainArgsK& A = *(ChainArgsK*)__builtin_amdgcn_kernarg_segment_ptr();
    extern __shared__ __attribute__((aligned(16))) unsigned char assoc_lds[];
    AssocArena* ar_ptr = reinterpret_cast<AssocArena*>(assoc_lds);
    asm volatile("" : "+s"(ar_ptr));     // (opaque: see chain_kernel)
    AssocArena& ar = *ar_ptr;
    __shared__ int s_task, s_nt;
    const int tid = threadIdx.x, wave = tid >> 6;
    unsigned* const err = A.flags + A.n_chains;
    unsigned* const tickets = A.flags + 2 * A.n_chains + 4;     // [0] kernel A's, [1] kernel B's
    unsigned* const assoc_done = tickets + 2;                   // [b]: frames of chain b associated
    const int n_tasks = A.n_chains * A.L;
    int done = 0;
    // the solver wave of the association shares its SIMD with a wave of kernel B: the association is the latency chain
    __builtin_amdgcn_s_setprio(1);
    while (true) {
        if (tid == 0) {
            int task = (int)__hip_atomic_fetch_add(tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (task >= n_tasks) task = -1;
            else {
                const int t = task / A.n_chains, b = task - t * A.n_chains;
                // the chain's tracklet table after frame t - 1 (kernel B)
                if (t > 0 && !split_wait(A.flags + b, (unsigned)t, err)) task = -1;
            }
            s_task = task;
        }
        __syncthreads();
        const int task = uni(s_task);
        if (task < 0) break;
        const int t = task / A.n_chains, b = task - t * A.n_chains, f = b * A.L + t;
        const long long c0 = clock64();
        if (tid == 0) s_nt = mvmc_ld_i32(A.n_tracks + b);
        __syncthreads();
        const int nt = uni(s_nt);
        long long c1;
        if (nt <= 0) {
            assoc_graph_spatial(ar, A, b, f, &done);
            __syncthreads();
            c1 = clock64();
            assoc_als_spatial(ar, A, b, f, &done);
        } else {
            assoc_graph_temporal(ar, A, b, f, &done);
            __syncthreads();
            c1 = clock64();
            assoc_als_temporal(ar, A, b, &done);
        }
        __syncthreads();
        const long long c2 = clock64();
        if (tid == 0) {
            const int it = nt <= 0 ? A.iters_sp[b] : A.iters_st[b];
            if (it < 0) {
                __hip_atomic_store(A.flags + A.n_chains + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicOr(A.flags + A.n_chains + 4 + b, 4u);
            }
            if (A.out_iters) A.out_iters[f] = it;
        }
        if (wave == 0) chain_assign(A, b, f, &done);
        if (done != 0) return;   // (never: the phases write 0)
        if (A.out_cycles && tid == 0) {   // (A(b, t) and B(b, t - 1 / t) never run at the same time: plain updates)
            double* oc = A.out_cycles + (size_t)b * 8;
            const long long c3 = clock64();
            oc[0] = (t ? oc[0] : 0.0) + (double)(c1 - c0);
            oc[1] = (t ? oc[1] : 0.0) + (double)(c2 - c1);
            oc[2] = (t ? oc[2] : 0.0) + (double)(c3 - c2);
            oc[6] = (t ? oc[6] : 0.0) + (double)(c3 - c0);
        }
        split_release(assoc_done + b, (unsigned)(t + 1));
        __syncthreads();      // (s_task / s_nt are rewritten by the next round)
    }
}

}  // namespace

int mvmc_chain_launch_assoc(const MvmcChainArgs& A, int n_blocks, hipStream_t stream) {
    if (hipFuncSetAttribute((const void*)chain_assoc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(AssocArena)) != hipSuccess)
        return MVMC_ERR_LAUNCH;
    hipLaunchKernelGGL(chain_assoc_kernel, dim3(n_blocks), dim3(512), sizeof(AssocArena), stream, A);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}
