// Kernel B of the split BIG layout (mvmc_chain_split.h): the IK problems of a chain-frame (four at a time, a wave each), commit and the
// per-frame outputs; persistent, one workgroup per CU beside kernel A's (three on a CU without one).  Its own translation unit: 168 VGPRs
// -- one wave per SIMD beside kernel A's two of 168 -- and the IK's batch sizes of the 168-register build (mvmc_common.h; the 128-register
// build measured 6 % slower here: 110.0 k against 115.3 k frames/s on config 5).
#define MVMC_SMALL_WPS 3
#define MVMC_NO_PHASE_PRIO
#define MVMC_CHAIN_SPLIT_TU
#include "mvmc_chain.hip"
#include "mvmc_chain_split.h"

namespace {

struct SolveArena { Ik1Shared ik[4]; int mq[64]; unsigned short mc[64]; };    // (POOL = 64: every pose of a C8 P8 frame)

__device__ __noinline__ void solve_ik(SolveArena& ar_in, const Ik1Tables& tables, ChainArgsK& A, int b, int* done) {
    SolveArena& ar = *uni(&ar_in);
    MVMC_ASSUME_LDS(&ar);
    MVMC_ASSUME_LDS(&tables);
    constexpr int NW = 4, POOL = 64;
    const int lane = threadIdx.x & 63;
    const int wave = uni((int)(threadIdx.x >> 6)), NP = A.T + A.K;
    const int cnt = lane < NP ? A.n_members[(size_t)b * NP + lane] : 0;
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
    const int excl = incl - cnt;
    // the frame's problem slots are handed out dynamically (a word in LDS): a slot without members costs a few instructions, a cold
    // solve several warm ones -- a static slot -> wave map left waves idle while the others worked through their share
    __shared__ int s_next;
    if (threadIdx.x == 0) s_next = NW;
    __syncthreads();
    for (int s = wave; s < NP; s = uni(lane == 0 ? atomicAdd(&s_next, 1) : 0)) {
        const int p = b * NP + s;
        const int base = uni(__shfl(excl, s, 64)), n_valid = uni(__shfl(cnt, s, 64));
        const int room = base < POOL ? POOL - base : 0;
        // (ik_scratch: eight blocks per chain, as chain_kernel<true> uses them; this kernel's four waves take the first four)
        ik1_solve(ar.ik[wave], ar.mq + (base < POOL ? base : 0), ar.mc + (base < POOL ? base : 0), room, tables, A.kps17, A.Pm, A.members,
                  p, A.V, A.C, A.P, A.init, A.cold, A.nfev_cold, A.nfev_warm, A.ik_params, A.ik_joints, A.ik_info,
                  A.ik_scratch + (ptrdiff_t)(b * 8 + wave - p) * MVMC_IK_SCRATCH_DOUBLES, 3, nullptr,
                  reinterpret_cast<int32_t*>(A.flags + A.n_chains + 4 + b), n_valid);
    }
    *done = 0;
}

__global__ void __launch_bounds__(256, 3)
chain_solve_kernel(Ik1Tables tables_arg, ChainArgs A_by_value) {
    constexpr size_t A_OFFSET = (sizeof(Ik1Tables) + alignof(ChainArgs) - 1) / alignof(ChainArgs) * alignof(ChainArgs);
    typedef const __attribute__((address_space(4))) char* KernargBytes;
    ChainArgsK& A = *(ChainArgsK*)((KernargBytes)__builtin_amdgcn_kernarg_segment_ptr() + A_OFFSET);
    extern __shared__ __attribute__((aligned(16))) unsigned char solve_lds[];
    SolveArena* ar_ptr = reinterpret_cast<SolveArena*>(solve_lds);
    asm volatile("" : "+s"(ar_ptr));
    SolveArena& ar = *ar_ptr;
    __shared__ Ik1Tables tables;
    __shared__ int s_task;
    const int tid = threadIdx.x, wave = tid >> 6;
    {
        typedef const __attribute__((address_space(4))) unsigned* KernargWords;
        KernargWords src = (KernargWords)__builtin_amdgcn_kernarg_segment_ptr();
        unsigned* dst = reinterpret_cast<unsigned*>(&tables);
        for (int i = tid; i < (int)(sizeof(Ik1Tables) / 4); i += 256) dst[i] = src[i];
    }
    unsigned* const err = A.flags + A.n_chains;
    unsigned* const tickets = A.flags + 2 * A.n_chains + 4;
    unsigned* const assoc_done = tickets + 2;
    const int n_tasks = A.n_chains * A.L, T = A.T, NP = T + A.K;
    int done = 0;
    __syncthreads();
    while (true) {
        if (tid == 0) {
            int task = (int)__hip_atomic_fetch_add(tickets + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (task >= n_tasks) task = -1;
            else {
                const int t = task / A.n_chains, b = task - t * A.n_chains;
                if (!split_wait(assoc_done + b, (unsigned)(t + 1), err)) task = -1;     // the frame's problem list (kernel A)
            }
            s_task = task;
        }
        __syncthreads();
        const int task = uni(s_task);
        if (task < 0) break;
        const int t = task / A.n_chains, b = task - t * A.n_chains, f = b * A.L + t;
        const long long c0 = clock64();
        solve_ik(ar, tables, A, b, &done);
        __syncthreads();
        const long long c1 = clock64();
        if (wave == 0) chain_commit(A, b, &done);
        __syncthreads();
        const long long c2 = clock64();
        for (int e = tid; e < T * 68; e += 256) A.out_params[(size_t)f * T * 68 + e] = A.params[(size_t)b * T * 68 + e];
        for (int e = tid; e < T * 54; e += 256) A.out_joints[(size_t)f * T * 54 + e] = A.joints[(size_t)b * T * 54 + e];
        for (int e = tid; e < T * 4; e += 256) A.out_meta[(size_t)f * T * 4 + e] = A.meta[(size_t)b * T * 4 + e];
        if (A.out_info)
            for (int e = tid; e < NP * 8; e += 256) A.out_info[(size_t)f * NP * 8 + e] = A.ik_info[(size_t)b * NP * 8 + e];
        if (tid == 0) {
            A.out_n[f] = mvmc_ld_i32(A.n_tracks + b);      // (the frame's association iterations: kernel A wrote out_iters)
            const unsigned v = __hip_atomic_load(A.flags + A.n_chains + 4 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 3u;
            if (v) atomicOr(A.flags + A.n_chains + 2, v);
        }
        if (done != 0) return;
        if (A.out_cycles && tid == 0) {
            double* oc = A.out_cycles + (size_t)b * 8;
            const long long c3 = clock64();
            oc[3] = (t ? oc[3] : 0.0) + (double)(c1 - c0);
            oc[4] = (t ? oc[4] : 0.0) + (double)(c2 - c1);
            oc[5] = (t ? oc[5] : 0.0) + (double)(c3 - c2);
            oc[6] += (double)(c3 - c0);
            oc[7] = (double)(t + 1);
        }
        split_release(A.flags + b, (unsigned)(t + 1));
        __syncthreads();
    }
}

}  // namespace

int mvmc_chain_launch_solve(const void* tables_host, const MvmcChainArgs& A, int n_blocks, hipStream_t stream) {
    if (hipFuncSetAttribute((const void*)chain_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SolveArena)) != hipSuccess)
        return MVMC_ERR_LAUNCH;
    hipLaunchKernelGGL(chain_solve_kernel, dim3(n_blocks), dim3(256), sizeof(SolveArena), stream, *static_cast<const Ik1Tables*>(tables_host), A);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}
