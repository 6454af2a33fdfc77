// Pieces shared by the IK kernels (mvmc_ik1.hip: one wave per solve; mvmc_ik_fd.hip: the TRF-faithful diagnostic solver).
#pragma once
#include "mvmc_common.h"

namespace {

constexpr int VMAX = 8;    // max views per person
constexpr int NOBS = 16;   // observed joints per view

// skeleton joint <-> observed keypoint (COCO-17 + synthetic mid-spine at 17); inverse_kinematics.py:366-378
#define MVMC_IK_SKEL_LIST {1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15, 16, 17}
__device__ __constant__ const int kIkSkel[NOBS] = MVMC_IK_SKEL_LIST;
__device__ __constant__ const int kIkObs[NOBS] = {11, 13, 15, 12, 14, 16, 17, 5, 7, 9, 6, 8, 10, 0, 3, 4};

// 4x4 symmetric Jacobi for the cold-start DLT of one joint (same scheme as mvmc_geom.hip)
template <int P, int Q>
__device__ __forceinline__ void rot4(double (&a)[4][4], double (&v)[4][4]) {
    const double apq = a[P][Q];
    if (fabs(apq) < 1e-300) return;
    const double theta = (a[Q][Q] - a[P][P]) / (2.0 * apq);
    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const double x = a[k][P], y = a[k][Q]; a[k][P] = c * x - s * y; a[k][Q] = s * x + c * y; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { const double x = a[P][k], y = a[Q][k]; a[P][k] = c * x - s * y; a[Q][k] = s * x + c * y; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { const double x = v[k][P], y = v[k][Q]; v[k][P] = c * x - s * y; v[k][Q] = s * x + c * y; }
}

// DLT of one observed keypoint (index into the 18-row pose: 17 = mid-spine) over the problem's views
__device__ void dlt_obs_point(const double* pose18 /*[V][18][3]*/, const double* Pm, int nv, int jo, double min_score,
                              double* X) {
    int n_ok = 0;
    for (int v = 0; v < nv; ++v) n_ok += pose18[(v * 18 + jo) * 3 + 2] >= min_score;
    const bool use_all = n_ok < 2;
    double a[4][4], vv[4][4];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { a[r][c] = 0.0; vv[r][c] = (r == c) ? 1.0 : 0.0; }
    for (int v = 0; v < nv; ++v) {
        const double* kp = &pose18[(v * 18 + jo) * 3];
        if (!use_all && !(kp[2] >= min_score)) continue;
        const double* P = &Pm[v * 12];
        double r1[4], r2[4];
        for (int k = 0; k < 4; ++k) { r1[k] = kp[0] * P[8 + k] - P[k]; r2[k] = kp[1] * P[8 + k] - P[4 + k]; }
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) a[r][c] += r1[r] * r1[c] + r2[r] * r2[c];
    }
    const double tr = a[0][0] + a[1][1] + a[2][2] + a[3][3];
    for (int sweep = 0; sweep < 16; ++sweep) {
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[0][3] * a[0][3] + a[1][2] * a[1][2] +
                           a[1][3] * a[1][3] + a[2][3] * a[2][3];
        if (off <= 1e-36 * tr * tr) break;
        rot4<0, 1>(a, vv); rot4<0, 2>(a, vv); rot4<0, 3>(a, vv); rot4<1, 2>(a, vv); rot4<1, 3>(a, vv); rot4<2, 3>(a, vv);
    }
    int m = 0;
    for (int k = 1; k < 4; ++k) if (a[k][k] < a[m][m]) m = k;
    double e[4];
    for (int r = 0; r < 4; ++r) e[r] = (m == 0) ? vv[r][0] : (m == 1) ? vv[r][1] : (m == 2) ? vv[r][2] : vv[r][3];
    X[0] = e[0] / e[3]; X[1] = e[1] / e[3]; X[2] = e[2] / e[3];
}

}  // namespace
