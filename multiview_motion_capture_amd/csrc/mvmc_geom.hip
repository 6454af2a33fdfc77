// Geometry kernels for gfx950: batched multi-view DLT (TR-1/TR-2) and forward kinematics (FK-1/FK-2).
#include "mvmc_common.h"
#include "mvmc_postopt.h"

// ------------------------------------------------------------------------------------------------
// DLT: one thread per (problem, joint).  The 2V x 4 system is reduced to its 4x4 normal matrix in
// fp64 registers and the null vector is the eigenvector of the smallest eigenvalue (== last right
// singular vector of A, mv_math_util.py:235-236); cyclic Jacobi, fully unrolled -> no scratch.
// ------------------------------------------------------------------------------------------------
template <int P, int Q>
__device__ __forceinline__ void jacobi_rot4(double (&a)[4][4], double (&v)[4][4]) {
    const double apq = a[P][Q];
    if (fabs(apq) < 1e-300) return;
    const double theta = (a[Q][Q] - a[P][P]) / (2.0 * apq);
    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double akp = a[k][P], akq = a[k][Q];
        a[k][P] = c * akp - s * akq;
        a[k][Q] = s * akp + c * akq;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double apk = a[P][k], aqk = a[Q][k];
        a[P][k] = c * apk - s * aqk;
        a[Q][k] = s * apk + c * aqk;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double vkp = v[k][P], vkq = v[k][Q];
        v[k][P] = c * vkp - s * vkq;
        v[k][Q] = s * vkp + c * vkq;
    }
}

__global__ void __launch_bounds__(256)
dlt_kernel(const double* __restrict__ kps17, const double* __restrict__ Pm, const int32_t* __restrict__ members,
           int B, int V, int C, int Pmax, int J, double min_score, double* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * J) return;
    const int b = idx / J, j = idx - b * J;
    const int ps = J * 3;  // pose stride
    const int32_t* mem = members + (size_t)b * V;
    int n_all = 0, n_ok = 0;
    for (int v = 0; v < V; ++v) {
        const int q = mem[v];
        if (q < 0) continue;
        ++n_all;
        if (kps17[(size_t)q * ps + j * 3 + 2] >= min_score) ++n_ok;
    }
    double* o = out + (size_t)idx * 4;
    if (n_all == 0) {
        const double nan = __longlong_as_double(0x7ff8000000000000LL);
        o[0] = o[1] = o[2] = o[3] = nan;
        return;
    }
    const bool use_all = n_ok < 2;  // "< 2 valid views -> resort to all views" (mv_math_util.py:177-182)
    double a[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) a[r][c] = 0.0;
    double ssum = 0.0;
    int nused = 0;
    for (int v = 0; v < V; ++v) {
        const int q = mem[v];
        if (q < 0) continue;
        const double* kp = kps17 + (size_t)q * ps + j * 3;
        const double x = kp[0], y = kp[1], sc = kp[2];
        if (!use_all && !(sc >= min_score)) continue;
        const double* Pc = Pm + (size_t)((q / Pmax) % C) * 12;
        double r1[4], r2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            r1[k] = x * Pc[8 + k] - Pc[k];
            r2[k] = y * Pc[8 + k] - Pc[4 + k];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) a[r][c] += r1[r] * r1[c] + r2[r] * r2[c];
        ssum += sc;
        ++nused;
    }
    double vv[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) vv[r][c] = (r == c) ? 1.0 : 0.0;
    const double tr = a[0][0] + a[1][1] + a[2][2] + a[3][3];
    for (int sweep = 0; sweep < 16; ++sweep) {
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[0][3] * a[0][3] + a[1][2] * a[1][2] +
                           a[1][3] * a[1][3] + a[2][3] * a[2][3];
        if (off <= 1e-36 * tr * tr) break;
        jacobi_rot4<0, 1>(a, vv); jacobi_rot4<0, 2>(a, vv); jacobi_rot4<0, 3>(a, vv);
        jacobi_rot4<1, 2>(a, vv); jacobi_rot4<1, 3>(a, vv); jacobi_rot4<2, 3>(a, vv);
    }
    double lmin = a[0][0];
    double e0 = vv[0][0], e1 = vv[1][0], e2 = vv[2][0], e3 = vv[3][0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (a[k][k] < lmin) { lmin = a[k][k]; e0 = vv[0][k]; e1 = vv[1][k]; e2 = vv[2][k]; e3 = vv[3][k]; }
    o[0] = e0 / e3; o[1] = e1 / e3; o[2] = e2 / e3;
    o[3] = ssum / (double)nused;
}

// ------------------------------------------------------------------------------------------------
// post-optimise of triangulated points (mv_math_util.py:189-210): one wave per problem
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
postopt_kernel(const double* __restrict__ kps, const double* __restrict__ Pm, const int32_t* __restrict__ members,
               int V, int C, int Pmax, int J, double* __restrict__ pts) {
    extern __shared__ double sm[];
    const int b = blockIdx.x, lane = threadIdx.x;
    double* sObs = sm;                 // [nv][J][3]
    double* sP = sm + V * J * 3;       // [nv][12]
    __shared__ int s_nv;
    __shared__ int s_q[64];
    if (lane == 0) {
        int nv = 0;
        for (int v = 0; v < V && nv < 64; ++v) {
            const int q = members[(size_t)b * V + v];
            if (q >= 0) s_q[nv++] = q;
        }
        s_nv = nv;
    }
    __syncthreads();
    const int nv = s_nv;
    if (nv < 1) return;
    for (int e = lane; e < nv * J * 3; e += 64) {
        const int v = e / (J * 3), r = e - v * J * 3;
        sObs[e] = kps[(size_t)s_q[v] * J * 3 + r];
    }
    for (int e = lane; e < nv * 12; e += 64) {
        const int v = e / 12, r = e - v * 12;
        sP[e] = Pm[(size_t)((s_q[v] / Pmax) % C) * 12 + r];
    }
    __syncthreads();
    double X[3] = {0, 0, 0};
    double* o = pts + ((size_t)b * J + (lane < J ? lane : 0)) * 4;
    if (lane < J) { X[0] = o[0]; X[1] = o[1]; X[2] = o[2]; }
    postopt::post_optimize_wave(X, sObs + (lane < J ? lane : 0) * 3, J * 3, sP, nv, J);
    if (lane < J) { o[0] = X[0]; o[1] = X[1]; o[2] = X[2]; }
}

// ------------------------------------------------------------------------------------------------
// FK: R_j = Rx Ry Rz through the reference's quaternion path (axis scaled by 1/(1+1e-10),
// Quaternions.py:444), chained 4x4 products in index order (parents precede children).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
fk_kernel(SkelDev sk, const double* __restrict__ params, int B, double* __restrict__ joints, double* __restrict__ Gout) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int np = 57 + sk.n_side;
    const double* x = params + (size_t)b * np;
    double Gm[18][12];  // rows 0..2 of each global 4x4 (row 3 is 0 0 0 1)
    for (int j = 0; j < 18; ++j) {
        double R[9], t[3];
        euler_to_rot(x + 3 + 3 * j, R);
        if (j == 0) {
            t[0] = x[0]; t[1] = x[1]; t[2] = x[2];
        } else {
            const double len = x[57 + sk.side_map[j]];
            t[0] = sk.dirs[j][0] * len; t[1] = sk.dirs[j][1] * len; t[2] = sk.dirs[j][2] * len;
        }
        if (j == 0) {
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) Gm[0][r * 4 + c] = R[r * 3 + c];
                Gm[0][r * 4 + 3] = t[r];
            }
        } else {
            const double* Gp = Gm[sk.parents[j]];
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c)
                    Gm[j][r * 4 + c] = Gp[r * 4] * R[c] + Gp[r * 4 + 1] * R[3 + c] + Gp[r * 4 + 2] * R[6 + c];
                Gm[j][r * 4 + 3] = Gp[r * 4] * t[0] + Gp[r * 4 + 1] * t[1] + Gp[r * 4 + 2] * t[2] + Gp[r * 4 + 3];
            }
        }
    }
    for (int j = 0; j < 18; ++j) {
        for (int r = 0; r < 3; ++r) joints[((size_t)b * 18 + j) * 3 + r] = Gm[j][r * 4 + 3];
        if (Gout) {
            double* g = Gout + ((size_t)b * 18 + j) * 16;
            for (int e = 0; e < 12; ++e) g[e] = Gm[j][e];
            g[12] = 0; g[13] = 0; g[14] = 0; g[15] = 1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" int mvmc_dlt(const double* kps17, const double* Pmats, const int32_t* members, int n_problems,
                        int v_max, int n_views, int p_max, int n_joints, double min_score, double* out,
                        mvmcStream_t stream) {
    if (!kps17 || !Pmats || !members || !out || v_max <= 0 || n_views <= 0 || p_max <= 0 || n_joints <= 0)
        return MVMC_ERR_ARG;
    if (n_problems <= 0) return n_problems == 0 ? MVMC_OK : MVMC_ERR_ARG;
    const int total = n_problems * n_joints;
    hipLaunchKernelGGL(dlt_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, kps17, Pmats, members,
                       n_problems, v_max, n_views, p_max, n_joints, min_score, out);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}

extern "C" int mvmc_fk(const mvmcSkeleton* skel_host, const double* params, int n_problems, double* joints,
                       double* G, mvmcStream_t stream) {
    if (!skel_host || !params || !joints) return MVMC_ERR_ARG;
    if (n_problems <= 0) return n_problems == 0 ? MVMC_OK : MVMC_ERR_ARG;
    SkelDev sk;
    if (!skel_to_dev(skel_host, &sk)) return MVMC_ERR_ARG;
    hipLaunchKernelGGL(fk_kernel, dim3((n_problems + 63) / 64), dim3(64), 0, (hipStream_t)stream, sk, params,
                       n_problems, joints, G);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}

extern "C" int mvmc_triangulate_postopt(const double* kps, const double* Pmats, const int32_t* members, int n_problems,
                                        int v_max, int n_views, int p_max, int n_joints, double* pts,
                                        mvmcStream_t stream) {
    if (!kps || !Pmats || !members || !pts || v_max <= 0 || v_max > 64 || n_views <= 0 || p_max <= 0) return MVMC_ERR_ARG;
    if (n_joints <= 0 || n_joints > 64) return MVMC_ERR_UNSUPPORTED;
    if (n_problems <= 0) return n_problems == 0 ? MVMC_OK : MVMC_ERR_ARG;
    const size_t shm = ((size_t)v_max * n_joints * 3 + (size_t)v_max * 12) * sizeof(double);
    if (shm > 60 * 1024) return MVMC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(postopt_kernel, dim3(n_problems), dim3(64), shm, (hipStream_t)stream, kps, Pmats, members, v_max,
                       n_views, p_max, n_joints, pts);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}
