// Diagnostic entry points (declared in include/mvmc.h under "diagnostics"): they run building blocks of the
// solver kernels on caller-supplied data so that tests can check them in isolation.
#include "mvmc_common.h"
#include "mvmc_eigh_tri.h"

namespace {
constexpr int NMAXE = 50, LDE = 51;

__global__ void __launch_bounds__(256)
debug_eigh_kernel(const double* __restrict__ Ain, const double* __restrict__ gin, int n, double* __restrict__ lam_out,
                  double* __restrict__ vt_out, int32_t* __restrict__ k0_out, double* __restrict__ cyc_out) {
    __shared__ double A[NMAXE * LDE], Z[NMAXE * LDE], W[NMAXE * LDE];
    __shared__ double g[64], lam[64], d[64], e[64], tau[64], pv[64], wv[64], red[8];
    __shared__ int icnt[1024];
    __shared__ long long prof[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int idx = tid; idx < n * n; idx += 256) A[(idx / n) * LDE + idx % n] = Ain[(size_t)b * n * n + idx];
    if (tid < n) g[tid] = gin[(size_t)b * n + tid];
    __syncthreads();
    const int k0 = eightri::eigh(A, LDE, Z, LDE, W, LDE, n, lam, d, e, tau, pv, wv, red, icnt, prof);
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += 256) vt_out[(size_t)b * n * n + idx] = Z[(idx / n) * LDE + idx % n];
    if (tid < n) lam_out[(size_t)b * n + tid] = lam[tid];
    if (tid == 0) k0_out[b] = k0;
    if (tid < 5 && cyc_out) cyc_out[(size_t)b * 5 + tid] = (double)prof[tid];
}
}  // namespace

extern "C" int mvmc_debug_eigh(const double* A, const double* g, int n_problems, int n, double* lam, double* Vt,
                               int32_t* k0, double* phase_cycles, mvmcStream_t stream) {
    if (!A || !g || !lam || !Vt || !k0 || n < 3 || n > NMAXE) return MVMC_ERR_ARG;
    if (n_problems <= 0) return n_problems == 0 ? MVMC_OK : MVMC_ERR_ARG;
    hipLaunchKernelGGL(debug_eigh_kernel, dim3(n_problems), dim3(256), 0, (hipStream_t)stream, A, g, n, lam, Vt, k0, phase_cycles);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}
