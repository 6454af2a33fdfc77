// The BIG layout (config 5, C8 P8) as TWO co-resident persistent kernels (round 5), shared definitions.
//
// chain_kernel<true> runs a chain-frame on one 512-thread workgroup per CU: graph -> association (52 % of its cycles, a latency chain on a
// solver wave with the vector pipes ~40 % busy) -> IK (30 %, eight waves, two per SIMD) -> commit.  Neither phase fills the CU, and a second
// such workgroup fits neither the registers (256 per lane) nor the LDS (129 KB).  Split by PHASE the two halves do fit one CU together:
//   A  chain_assoc_kernel (mvmc_chain_assoc.hip): 512 threads, <= 168 VGPRs (two waves per SIMD), 111 KB of LDS (the symmetrised affinity
//      of als5 in a global buffer instead of LDS) -- graph, association, assignment of one chain-frame;
//   B  chain_solve_kernel (mvmc_chain_solve.hip): 256 threads, <= 168 VGPRs (one wave per SIMD beside A's two), 42 KB of LDS -- the frame's
//      IK problems four at a time, commit, the per-frame outputs.
// Both are PERSISTENT (one workgroup per CU each) and draw chain-frames in ticket order (frame-major: ticket = t * n_chains + chain), so a
// CU associates one chain while it solves another.  Per chain the two alternate strictly -- A(b, t) waits for B(b, t - 1)'s tracklet table,
// B(b, t) for A(b, t)'s problem list -- through two words per chain (flags[b] = frames solved, ring[b] = frames associated), producer:
// vmcnt(0) + barrier + agent-scope release + relaxed store; consumer: one lane polls relaxed, agent-scope acquire, barrier.  Every wait is
// bounded by wall time (4 s) and ends on the launch's error word.  No deadlock: tickets are taken by RUNNING workgroups in order, so the
// lowest outstanding task of either kind has its predecessor finished or running, as long as one workgroup of each kernel is resident --
// A is launched first and cannot share a CU with another A (LDS), B fits beside it (registers and LDS counted above).
// The device functions are those of chain_kernel<true>; results are bit-identical (tests/test_gpu_config5_c8p8.py).
//
// STATUS: an experiment, off by default (run_chains_fused(split=True), bench.py --big-split).  Measured on config 5 (25,008 frames):
// 115 - 119 k frames/s against 108 k for chain_kernel<true> (+ 7 - 10 %) -- kernel B is the longer stage (a CU's registers leave room for
// four IK waves beside the association's eight, so a frame's eight solves take two rounds).  HAZARD: A and B wait for each other, so they
// must run CONCURRENTLY -- i.e. sit in different hardware queues.  HIP multiplexes a process's streams onto a few hardware queues; when
// the two kernels (or those of two steps in flight) share one, the queue runs them in order and the bounded wait times out (loud, 4 s;
// seen with bench.py's two steps in flight).  A caller that uses it keeps the number of live streams small.
#pragma once

// lane 0 of a workgroup: wait until *word >= need; false = the launch's error word is set (by this wait's time-out or another's)
__device__ __forceinline__ bool split_wait(const unsigned* word, unsigned need, unsigned* err) {
    const unsigned long long t0 = wall_clock64();
    unsigned spins = 0;
    bool ok = true;
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(32);
        if ((++spins & 1023u) != 0u) continue;
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = false; break; }
        if (wall_clock64() - t0 > 400000000ull) {   // ~4 s at the 100 MHz constant clock
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = false;
            break;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return ok;
}
// every thread of the workgroup calls it after the task's last store: the task's results are visible at agent scope before the word moves
__device__ __forceinline__ void split_release(unsigned* word, unsigned value) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
