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
    hipMalloc(&out, sizeof(double) * 256 * cus * 16);
    printf("device %s, %d CUs, clock %d MHz\n", p.gcnArchName, cus, p.clockRate / 1000);
    printf("%-28s %10s %10s %8s\n", "waves per SIMD", "MFMA f64", "FMA f64", "ratio");
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = cus * wps;   // 256 threads = 4 waves = one per SIMD, wps workgroups per CU
        const double mf = run(k_mfma, blocks, out, (double)ITERS * NCHAIN * 2048.0);
        const double ff = run(k_fma, blocks, out, (double)ITERS * NCHAIN * 4 * 128.0);
        printf("%-28d %8.1f TF %8.1f TF %8.2f\n", wps, mf, ff, mf / ff);
    }
    printf("(dense fp64 peaks of MI355X: vector 78.6 TFLOP/s, matrix 78.6 TFLOP/s -- MI355X_MICROARCH.md; an MFMA instruction is 2,048 flop, "
           "an FMA wave instruction 128)\n");
    hipFree(out);
    return 0;
}
