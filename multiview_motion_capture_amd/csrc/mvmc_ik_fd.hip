// TRF-faithful IK on the device (diagnostic; include/mvmc.h: mvmc_debug_ik_solve_fd).
//
// PoseSolver.solve (inverse_kinematics.py:380-433) with the reference's own numerical method instead of the production solver's:
// 2-point finite-difference Jacobians and the SVD-based trust-region step of scipy.optimize.least_squares, restated in
// mvmc_trf_faithful.h.  One 64-lane wave per solve, matrices in a caller-supplied global workspace, no tuning: it exists to measure
// how far an independent implementation of the REFERENCE'S algorithm lands from the reference's truncated solves (the band inside
// which the production kernel, mvmc_ik1.hip, is then judged: tests/test_gpu_ik.py, DESIGN.md "IK parity").
#include "mvmc_common.h"
#include "mvmc_trf_faithful.h"

namespace {

namespace tf = trf_faithful;

__global__ void __launch_bounds__(64)
ik_fd_kernel(SkelDev skarg, const double* __restrict__ kps17, const double* __restrict__ Pmats, const int32_t* __restrict__ members,
             int V, int C, int Pmax, const double* __restrict__ init, const uint8_t* __restrict__ cold, int nfev_cold, int nfev_warm,
             int stage_mask, double* __restrict__ params_out, double* __restrict__ joints_out, double* __restrict__ info_out,
             double* __restrict__ work_all) {
    constexpr int VM = 8;
    __shared__ double pose18[VM * 54], Pm[VM * 12], x[68], side0[18];
    __shared__ tf::Skel sk;
    __shared__ int s_nv;
    const int b = blockIdx.x, lane = threadIdx.x;
    const tf::Wave64 ex;
    double* work = work_all + (size_t)b * MVMC_IK_FD_WORK_DOUBLES;
    if (lane < 18) {
        for (int k = 0; k < 3; ++k) sk.dirs[lane][k] = skarg.dirs[lane][k];
        sk.parents[lane] = skarg.parents[lane];
        sk.side_map[lane] = skarg.side_map[lane];
    }
    if (lane == 0) {
        sk.n_side = skarg.n_side;
        int nv = 0;
        for (int v = 0; v < V && nv < VM; ++v) {
            const int q = members[(size_t)b * V + v];
            if (q < 0) continue;
            const double* kp = kps17 + (size_t)q * 51;
            double* dst = pose18 + nv * 54;
            for (int e = 0; e < 51; ++e) dst[e] = kp[e];
            // synthetic mid-spine row (inverse_kinematics.py:339-348)
            for (int c = 0; c < 2; ++c) {
                const double mid_sh = 0.5 * (kp[5 * 3 + c] + kp[6 * 3 + c]);
                const double mid_hip = 0.5 * (kp[11 * 3 + c] + kp[12 * 3 + c]);
                dst[51 + c] = 0.5 * (mid_sh + mid_hip);
            }
            double sc = kp[5 * 3 + 2] * kp[6 * 3 + 2];
            sc *= kp[11 * 3 + 2] * kp[12 * 3 + 2];
            dst[53] = sc;
            const double* Pc = Pmats + (size_t)((q / Pmax) % C) * 12;
            for (int e = 0; e < 12; ++e) Pm[nv * 12 + e] = Pc[e];
            ++nv;
        }
        s_nv = nv;
    }
    __syncthreads();
    const int nv = s_nv, n_side = skarg.n_side;
    double* info = info_out ? info_out + (size_t)b * 8 : nullptr;
    if (nv < 2) {   // the reference only solves clusters with >= 2 views (motion_capture.py:927,940)
        const double nan = __longlong_as_double(0x7ff8000000000000LL);
        for (int i = lane; i < 68; i += 64) params_out[(size_t)b * 68 + i] = nan;
        if (lane < 54) joints_out[(size_t)b * 54 + lane] = nan;
        if (info && lane < 8) info[lane] = nan;
        return;
    }
    const bool is_cold = cold == nullptr || cold[b] != 0;
    if (is_cold) {
        double* p3d = work + tf::work_doubles(18 * nv, 54);   // behind the solver's own workspace
        tf::triangulate_postopt18(ex, pose18, Pm, nv, p3d, work);
        __syncthreads();
        if (lane < 3) x[lane] = 0.5 * (p3d[11 * 3 + lane] + p3d[12 * 3 + lane]);
        if (lane < 54) x[3 + lane] = 0.0;
        if (lane < n_side) x[57 + lane] = skarg.ref_side[lane];
    } else {
        for (int i = lane; i < 57 + n_side; i += 64) x[i] = init[(size_t)b * 68 + i];
    }
    __syncthreads();
    if (lane < n_side) side0[lane] = x[57 + lane];
    __syncthreads();
    const int max_nfev = is_cold ? nfev_cold : nfev_warm;
    tf::Result r[2] = {{0.0, 0, 0, 0}, {0.0, 0, 0, 0}};
    for (int stage = 0; stage < 2; ++stage) {
        if (!((stage_mask >> stage) & 1)) continue;
        tf::IkResidual fun{&sk, pose18, Pm, side0, nv, stage};
        r[stage] = tf::trf(ex, fun, stage == 0 ? 57 : 57 + n_side, fun.m(), x, max_nfev, work);
        __syncthreads();
    }
    for (int i = lane; i < 57 + n_side; i += 64) params_out[(size_t)b * 68 + i] = x[i];
    if (lane == 0) {
        double pos[54];
        tf::forward_kinematics(sk, x, x + 3, x + 57, pos);
        for (int e = 0; e < 54; ++e) joints_out[(size_t)b * 54 + e] = pos[e];
        if (info) {
            info[0] = r[0].cost; info[1] = r[0].nfev; info[2] = r[0].status;
            info[3] = r[1].cost; info[4] = r[1].nfev; info[5] = r[1].status;
            info[6] = r[0].njev + r[1].njev; info[7] = 0.0;
        }
    }
}

}  // namespace

extern "C" int mvmc_debug_ik_solve_fd(const mvmcSkeleton* skel_host, const double* kps17, const double* Pmats, const int32_t* members,
                                      int n_problems, int v_max, int n_views, int p_max, const double* init_params,
                                      const uint8_t* cold, int max_nfev_cold, int max_nfev_warm, int stage_mask, double* params_out,
                                      double* joints_out, double* info_out, double* work, mvmcStream_t stream) {
    if (!skel_host || !kps17 || !Pmats || !members || !params_out || !joints_out || !work) return MVMC_ERR_ARG;
    if (v_max <= 0 || n_views <= 0 || p_max <= 0 || max_nfev_cold < 1 || max_nfev_warm < 1) return MVMC_ERR_ARG;
    if (stage_mask < 1 || stage_mask > 3) return MVMC_ERR_ARG;
    if (cold && !init_params) return MVMC_ERR_ARG;
    if (n_problems <= 0) return n_problems == 0 ? MVMC_OK : MVMC_ERR_ARG;
    SkelDev sk;
    if (!skel_to_dev(skel_host, &sk)) return MVMC_ERR_ARG;
    if (sk.n_side != MVMC_N_SIDE) return MVMC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(ik_fd_kernel, dim3(n_problems), dim3(64), 0, (hipStream_t)stream, sk, kps17, Pmats, members, v_max, n_views, p_max,
                       init_params, init_params ? cold : nullptr, max_nfev_cold, max_nfev_warm, stage_mask, params_out, joints_out,
                       info_out, work);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}
