// Layout probe for global_load_lds_dwordx3 on gfx950 (run on the GPU box): hipcc --offload-arch=gfx950 -O3 tools/glds12_test.hip -o /tmp/glds12 && /tmp/glds12
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* __restrict__ src, float* __restrict__ dst) {
    __shared__ float buf[1024];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 1024; i += 256) buf[i] = -1.0f;
    __syncthreads();
    // lane gathers source triple (w * 64 + lane) * 2 -> where does it land?
    const float* g = src + (size_t)(w * 64 + lane) * 2 * 3;
    float* l = buf + w * 64 * 3;   // wave-uniform
    if (lane != 5)    // a hole: lane 5 is switched off
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 12, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    for (int i = threadIdx.x; i < 1024; i += 256) dst[i] = buf[i];
}
int main() {
    const int n = 256 * 2 * 3 + 16;
    float *s, *d;
    (void)hipMalloc(&s, n * 4); (void)hipMalloc(&d, 1024 * 4);
    static float h[n]; for (int i = 0; i < n; ++i) h[i] = (float)i;
    (void)hipMemcpy(s, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, s, d);
    static float o[1024]; (void)hipMemcpy(o, d, 1024 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) for (int c = 0; c < 3; ++c) { const float want = (t & 63) == 5 ? -1.0f : (float)(t * 6 + c); if (o[t * 3 + c] != want) ++bad; }
    printf("glds12 gather, lane-linear 12-byte layout: %s (%d bad)\n", bad ? "NO" : "yes", bad);
    printf("first 48 floats of wave 0's block:"); for (int i = 0; i < 48; ++i) printf(" %g", o[i]); printf("\n");
    printf("floats 186..200:"); for (int i = 186; i < 200; ++i) printf(" %g", o[i]); printf("\n");
    return 0;
}
