// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 against v_fma_f64 on gfx950 (north star: "MFMA only for the dense J^T J").
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_f64_rate tools/mfma_f64_rate.hip && tools/bin/mfma_f64_rate
// Every wave runs ITERS iterations of NCHAIN independent accumulator chains (so the rate, not the latency of one chain, is measured);
// flops are counted as the instruction defines them: MFMA 16x16x4 = 2 * 16 * 16 * 4 = 2,048 per wave instruction, FMA = 2 * 64.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int ITERS = 4096, NCHAIN = 8;
typedef double double4v __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_mfma(double* out, double a0, double b0) {
    double4v acc[NCHAIN];
    for (int c = 0; c < NCHAIN; ++c) acc[c] = double4v{0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < NCHAIN; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    double s = 0.0;
    for (int c = 0; c < NCHAIN; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_fma(double* out, double a0, double b0) {
    double acc[NCHAIN * 4];
    for (int c = 0; c < NCHAIN * 4; ++c) acc[c] = c * 1e-3;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < NCHAIN * 4; ++c) acc[c] = __builtin_fma(a, acc[c], b);
    }
    double s = 0.0;
    for (int c = 0; c < NCHAIN * 4; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Mixed: 512-thread workgroups; waves 0 .. 3 (one per SIMD) run the MFMA loop, waves 4 .. 7 (their SIMD partners) the FMA loop (MODE 0),
// a loop of 32-bit integer VALU work (MODE 1) or nothing (MODE 2: calibration).  Do the matrix cores and the vector pipe run side by side?
template <int MODE>
__global__ void __launch_bounds__(512) k_mix(double* out, double a0, double b0) {
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    double s = 0.0;
    if (threadIdx.x < 256) {
        double4v acc[NCHAIN];
        for (int c = 0; c < NCHAIN; ++c) acc[c] = double4v{0.0, 0.0, 0.0, 0.0};
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int c = 0; c < NCHAIN; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
        }
        for (int c = 0; c < NCHAIN; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    } else if (MODE == 0) {
        double acc[NCHAIN * 4];
        for (int c = 0; c < NCHAIN * 4; ++c) acc[c] = c * 1e-3;
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int c = 0; c < NCHAIN * 4; ++c) acc[c] = __builtin_fma(a, acc[c], b);
        }
        for (int c = 0; c < NCHAIN * 4; ++c) s += acc[c];
    } else if (MODE == 2) {
        s = a;                                                                               // the partner exits at once: calibration
    } else {
        unsigned acc[NCHAIN * 4];
        const unsigned m = (unsigned)threadIdx.x * 2654435761u + 12345u;
        for (int c = 0; c < NCHAIN * 4; ++c) acc[c] = c + threadIdx.x;
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int c = 0; c < NCHAIN * 4; ++c) acc[c] = (acc[c] ^ m) + (acc[c] >> 3);      // three 32-bit VALU instructions
        }
        unsigned t = 0;
        for (int c = 0; c < NCHAIN * 4; ++c) t += acc[c];
        s = (double)t;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static double run_ms(K kernel, int blocks, double* out, int threads = 256) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, out, 1.0000001, 1e-9);   // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, out, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <typename K>
static double run(K kernel, int blocks, double* out, double flop_per_thread_iter_wave) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9);   // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = blocks * 4.0;
    return waves * flop_per_thread_iter_wave / (ms * 1e-3) / 1e12;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    double* out;
    hipMalloc(&out, sizeof(double) * 512 * cus * 16);
    printf("device %s, %d CUs, clock %d MHz\n", p.gcnArchName, cus, p.clockRate / 1000);
    printf("%-28s %10s %10s %8s\n", "waves per SIMD", "MFMA f64", "FMA f64", "ratio");
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = cus * wps;   // 256 threads = 4 waves = one per SIMD, wps workgroups per CU
        const double mf = run(k_mfma, blocks, out, (double)ITERS * NCHAIN * 2048.0);
        const double ff = run(k_fma, blocks, out, (double)ITERS * NCHAIN * 4 * 128.0);
        printf("%-28d %8.1f TF %8.1f TF %8.2f\n", wps, mf, ff, mf / ff);
    }
    printf("\nmixed: half of the waves MFMA f64, their SIMD partners vector work (ms for the same per-wave work as above)\n");
    printf("%-28s %12s %12s %12s %12s %12s\n", "waves per SIMD", "MFMA alone", "FMA alone", "MFMA+exit", "MFMA+FMA", "MFMA+int");
    for (int wps = 2; wps <= 8; wps *= 2) {
        const int blocks = cus * wps;
        // "alone" = the same number of workgroups of that kind as in the mixed launch (half of the blocks)
        const double m_alone = run_ms(k_mfma, blocks / 2, out), f_alone = run_ms(k_fma, blocks / 2, out);
        // int alone: the mixed kernel's odd branch only -- launch the mixed kernel with MODE 1 on a grid whose even blocks exit at once is
        // not expressible; measure it through the difference instead: k_mix<1> against MFMA alone
        const double mix_f = run_ms(k_mix<0>, blocks / 2, out, 512), mix_i = run_ms(k_mix<1>, blocks / 2, out, 512), mix_0 = run_ms(k_mix<2>, blocks / 2, out, 512);
        printf("%-28d %9.3f ms %9.3f ms %9.3f ms %9.3f ms %9.3f ms\n", wps, m_alone, f_alone, mix_0, mix_f, mix_i);
    }
    printf("(separate pipes: mixed = max of the two; one shared pipe: mixed = sum)\n");
    printf("(dense fp64 peaks of MI355X: vector 78.6 TFLOP/s, matrix 78.6 TFLOP/s -- MI355X_MICROARCH.md; an MFMA instruction is 2,048 flop, "
           "an FMA wave instruction 128)\n");
    hipFree(out);
    return 0;
}
