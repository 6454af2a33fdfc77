// Which lane holds which element in v_mfma_f64_16x16x4_f64 on gfx950?  D = A (16x4) B (4x16): prints the layout that matches.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/mfma_layout_test tools/mfma_layout_test.hip && tools/bin/mfma_layout_test
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4v __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* D) {   // A[i*4+k], B[k*16+j] row-major
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + l / 16];      // assumed: lane -> A[i = l % 16][k = l / 16]
    const double b = B[(l / 16) * 16 + l % 16];     // assumed: lane -> B[k = l / 16][j = l % 16]
    double4v c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[l * 4 + v] = c[v];
}
int main() {
    double hA[64], hB[64], hD[256], ref[16][16];
    for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 4; ++kk) hA[i * 4 + kk] = 1.0 + i + 0.01 * kk;
    for (int kk = 0; kk < 4; ++kk) for (int j = 0; j < 16; ++j) hB[kk * 16 + j] = 0.5 + 0.1 * j - 0.003 * kk * j;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int kk = 0; kk < 4; ++kk) s += hA[i * 4 + kk] * hB[kk * 16 + j]; ref[i][j] = s; }
    double *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    // candidate D layouts
    int ok1 = 1, ok2 = 1, ok3 = 1;
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
        const double d = hD[l * 4 + v];
        if (fabs(d - ref[4 * (l / 16) + v][l % 16]) > 1e-12) ok1 = 0;     // i = 4 (l / 16) + v, j = l % 16
        if (fabs(d - ref[(l / 16) + 4 * v][l % 16]) > 1e-12) ok2 = 0;     // i = l / 16 + 4 v,  j = l % 16
        if (fabs(d - ref[l % 16][4 * (l / 16) + v]) > 1e-12) ok3 = 0;     // transposed
    }
    printf("A[i = l %% 16][k = l / 16], B[k = l / 16][j = l %% 16];  D[4 (l / 16) + v][l %% 16]: %s;  D[l / 16 + 4 v][l %% 16]: %s;  D[l %% 16][4 (l / 16) + v]: %s\n",
           ok1 ? "MATCH" : "no", ok2 ? "MATCH" : "no", ok3 ? "MATCH" : "no");
    if (!ok1 && !ok2 && !ok3) for (int l = 0; l < 64; l += 17) printf("lane %d: %g %g %g %g (ref row %d: %g %g)\n", l, hD[l*4], hD[l*4+1], hD[l*4+2], hD[l*4+3], l % 16, ref[l%16][0], ref[l%16][1]);
    return 0;
}
