// Calibration of rocprofv3's FETCH_SIZE for the access patterns of the triangulation kernels (the x2 correction of a
// streaming read must be calibrated, not argued):  hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -- /tmp/fetch_calib
// Three kernels over the same 1.2 GB buffer of "poses" (25 joints x 12 bytes), each launched 3 times:
//   calib_copy16     every byte once, 16 bytes per lane, coalesced                      -> 1.2 GB of distinct bytes
//   calib_copy12     every byte once, 12 bytes per lane (x, y, score), coalesced        -> 1.2 GB
//   calib_gather_dma 17 of 25 joints per pose by global_load_lds_dwordx3 (the pattern of ingest_dlt3_kernel: lane = (pose, COCO joint))
//                    -> 0.816 GB requested, every 64-byte line of the buffer touched (1.2 GB from HBM)
// tools/aggregate_profiles.py divides the known bytes by the counter and applies that factor to the kernel with the same pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr size_t POSES = 4000000;            // x 300 B = 1.2 GB
__global__ void __launch_bounds__(256) calib_copy16(const float4* __restrict__ src, float* __restrict__ out, size_t n16) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void __launch_bounds__(256) calib_copy12(const float* __restrict__ src, float* __restrict__ out, size_t n12) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n12; i += (size_t)gridDim.x * 256) { const float* p = src + i * 3; acc += p[0] + p[1] + p[2]; }
    if (acc == 12345.678f) out[0] = acc;
}
__device__ __forceinline__ int op25_to_coco17(int j) {
    return j < 12 ? (int)((0x610e3308b193e00ull >> (5 * j)) & 31u) : (int)((0xb729a9u >> (5 * (j - 12))) & 31u);
}
__global__ void __launch_bounds__(256) calib_gather_dma(const float* __restrict__ src, float* __restrict__ out, size_t n_tr) {
    __shared__ float buf[256 * 4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float acc = 0.f;
    for (size_t t0 = blockIdx.x * 256ull; t0 < n_tr; t0 += (size_t)gridDim.x * 256) {
        const size_t t = t0 + threadIdx.x;
        if (t < n_tr) {
            const size_t q = t / 17; const int j = (int)(t - q * 17);
            const float* g = src + (q * 25 + op25_to_coco17(j)) * 3;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(buf + w * 256), 12, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += buf[w * 256 + lane * 4];
    }
    if (acc == 12345.678f) out[0] = acc;
}
int main() {
    float *src, *out;
    if (hipMalloc(&src, POSES * 300) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    (void)hipMemset(src, 0, POSES * 300);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_copy16, dim3(4096), dim3(256), 0, 0, (const float4*)src, out, POSES * 300 / 16);
        hipLaunchKernelGGL(calib_copy12, dim3(4096), dim3(256), 0, 0, src, out, POSES * 25);
        hipLaunchKernelGGL(calib_gather_dma, dim3(4096), dim3(256), 0, 0, src, out, POSES * 17);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("known bytes: calib_copy16 %zu calib_copy12 %zu calib_gather_dma %zu (requested %zu)\n", POSES * 300, POSES * 300, POSES * 300, POSES * 17 * 12);
    return 0;
}
