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

// One trust-region step in the Krylov tridiagonal basis (mvmc_eigh_tri.h: tridiag_krylov + krylov_block_ok +
// tr_solve_tri + apply_q_krylov), strung together as in the IK kernel: J = Bm (m x n), g = Bm^T r, M = Bm^T Bm.
__global__ void __launch_bounds__(256)
debug_trstep_kernel(const double* __restrict__ Bin, const double* __restrict__ rin, int m, int n, double Delta,
                    double alpha0, double* __restrict__ step_out, double* __restrict__ out4, double* __restrict__ cyc_out) {
    __shared__ double Bm[NMAXE * LDE], V[NMAXE * LDE];
    __shared__ double r[64], g[64], d[64], e[64], tau[64], rh[64], v0[64], red[8];
    __shared__ double lmul[64], dinv[64], yb[64], zb[64], cv[64], wn[64], dsc[64], e2sc[64], sc[8], k4[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int idx = tid; idx < m * n; idx += 256) Bm[(idx / n) * LDE + idx % n] = Bin[(size_t)b * m * n + idx];
    if (tid < m) r[tid] = rin[(size_t)b * m + tid];
    __syncthreads();
    if (tid < n) {
        double acc = 0.0;
        for (int i = 0; i < m; ++i) acc += Bm[i * LDE + tid] * r[i];
        g[tid] = acc;
    }
    __syncthreads();
    double gg = 0.0;
    for (int i = 0; i < n; ++i) gg += g[i] * g[i];
    double a[eightri::KQ];
    {
        const int w = tid >> 6, l = tid & 63;
#pragma unroll
        for (int q = 0; q < eightri::KQ; ++q) {
            const int i = w + 4 * q;
            double acc = 0.0;
            if (i < n && l < n)
                for (int c = 0; c < m; ++c) acc += Bm[c * LDE + i] * Bm[c * LDE + l];
            a[q] = acc;
        }
    }
    __shared__ long long prof[4];
    __shared__ double part[256], svx[66];
    const long long t_all = clock64();
    const int kk = eightri::tridiag_krylov(a, g, V, n, d, e, tau, v0, svx, part, red, k4, prof);
    const long long t_tri = clock64() - t_all;
    if (tid < 64) {
        double alpha = -1.0, pred = 0.0, pnorm = 0.0, cj = 0.0;
        const bool ok = kk > 0 && eightri::krylov_block_ok(d, e, kk, n, k4[2], k4[3], dsc, e2sc, lmul, dinv, wn);
        if (ok) {
            rh[tid] = tid == 0 ? k4[0] : 0.0;
            const double pivmin = 1e-16 * k4[2] + 1e-300;
            alpha = eightri::tr_solve_tri<false>(d, e, rh, kk, Delta, alpha0, gg, pivmin, lmul, dinv, yb, zb, cv, &pred, &pnorm);
            double c = tid < kk ? cv[tid] : 0.0;
            if (kk < n) {
                const double eta = k4[3] * wave_sum_dpp(tid < kk ? wn[tid] * c : 0.0);
                if (tid == kk) c = eta;
            }
            cj = eightri::apply_q_krylov(V, tau, v0, k4[1], kk, n, c);
        }
        cv[tid] = cj;
        if (tid == 0) { sc[0] = alpha; sc[1] = pred; sc[2] = pnorm; sc[3] = ok ? (double)kk : -1.0; }
    }
    __syncthreads();
    if (tid < n) step_out[(size_t)b * n + tid] = cv[tid];
    if (tid < 4) out4[(size_t)b * 4 + tid] = sc[tid];
    if (cyc_out && tid < 3) cyc_out[(size_t)b * 4 + tid] = (double)prof[tid];
    if (cyc_out && tid == 3) cyc_out[(size_t)b * 4 + 3] = (double)t_tri;
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

extern "C" int mvmc_debug_trstep(const double* B, const double* r, int n_problems, int m, int n, double Delta,
                                 double alpha0, double* step, double* out4, double* phase_cycles, mvmcStream_t stream) {
    if (!B || !r || !step || !out4 || m < 3 || n < 3 || m > NMAXE || n > NMAXE || !(Delta > 0.0)) return MVMC_ERR_ARG;
    if (n_problems <= 0) return n_problems == 0 ? MVMC_OK : MVMC_ERR_ARG;
    hipLaunchKernelGGL(debug_trstep_kernel, dim3(n_problems), dim3(256), 0, (hipStream_t)stream, B, r, m, n, Delta,
                       alpha0, step, out4, phase_cycles);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}
