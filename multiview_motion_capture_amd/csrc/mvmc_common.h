// Shared device/host helpers of the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mvmc.h"

#define MVMC_WAVE 64

// Waves per SIMD a translation unit's register-critical device functions (IK model, tridiagonalisation, als7) are built for.
// 4 = 128 VGPRs: the SMALL layout of the chain kernel since round 5 (mvmc_chain.hip sets it for its own unit) -- FOUR workgroups per
// CU: the arena is 40,944 B since the IK's observations left LDS, and the loops have batch sizes for the smaller file (MVMC_TRI_HB,
// MVMC_IK_GR, MVMC_IK_PARK below; the als7 loop in its role-split form).  3 = 168 VGPRs, the larger batches: the stand-alone kernels
// (one solve's or one graph's latency matters there, not the occupancy) and round 4's chain kernel (-DMVMC_SMALL_WPS=3 on the command
// line builds it again; DESIGN.md section 6a has the same-box comparison).  The knobs can be overridden one by one for experiments
// (tools/full_variant.sh).
#ifndef MVMC_SMALL_WPS
#define MVMC_SMALL_WPS 3
#endif
#if MVMC_SMALL_WPS >= 4
#ifndef MVMC_TRI_HB
#define MVMC_TRI_HB 3      // pairs of rows per LDS round trip of the tridiagonalisation's rank-2 update (5 in the 168-register build)
#endif
#ifndef MVMC_IK_GR
#define MVMC_IK_GR 2       // rows per chunk of the J^T J accumulation (4 in the 168-register build)
#endif
#ifndef MVMC_IK_PARK
#define MVMC_IK_PARK 1     // the lane's angular velocities wait in LDS during the accumulation
#endif
#endif

// Ordering point for LDS traffic INSIDE one wave (single-wave routines that may run in a multi-wave workgroup, where
// __syncthreads() would be a real barrier across waves doing unrelated work).  LDS operations of a wave execute in
// order, so only the compiler has to be kept from moving accesses across the point.
#define MVMC_WAVE_SYNC()                                        \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
    } while (0)

// For out-of-line device functions that receive LDS through generic pointers: tells the compiler that p is LDS, so that
// the accesses through it become ds_ instructions instead of flat_ ones (and keep their 32-bit addresses).
#if defined(__HIP_DEVICE_COMPILE__)
#define MVMC_ASSUME_LDS(p) __builtin_assume(__builtin_amdgcn_is_shared((const void*)(p)))
#else
#define MVMC_ASSUME_LDS(p) ((void)0)
#endif

// A double in global memory, for pointers that travel through out-of-line functions next to LDS traffic: accesses through a
// generic pointer are flat_ instructions, which count on the LDS counter too (a wait for an LDS result then also waits for the
// global access); through this type they are global_ instructions.
typedef __attribute__((address_space(1))) double mvmc_gdouble;

// Scalar state that another workgroup may have written earlier in the same launch (chain kernel, parts > 1): read on the
// vector path with an agent-scope load, never through the scalar cache (which an acquire fence does not refresh).
__device__ __forceinline__ int32_t mvmc_ld_i32(const int32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#define MVMC_CHECK_LAUNCH()                                  \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return MVMC_ERR_LAUNCH; \
    } while (0)

// OpenPose-25 row of each COCO-17 joint (pose_def.py:72-96 vs :111-137).
__device__ __constant__ const int kOp25ToCoco17[17] = {0, 16, 15, 18, 17, 5, 2, 6, 3, 7, 4, 12, 9, 13, 10, 14, 11};
// the same table in registers (five bits per joint): a lookup in the constant above is a dependent global load in front of every keypoint load
__device__ __forceinline__ int op25_to_coco17(int j) {
    return j < 12 ? (int)((0x610e3308b193e00ull >> (5 * j)) & 31u) : (int)((0xb729a9u >> (5 * (j - 12))) & 31u);
}

// Non-contracted double ops: used wherever the reference's NumPy expression rounds
// after every multiply/add and the result feeds a float32 store (bit-exact D / S).
__device__ __forceinline__ double dmul(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double dadd(double a, double b) { return __dadd_rn(a, b); }
__device__ __forceinline__ float fmulr(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float faddr(float a, float b) { return __fadd_rn(a, b); }
// NumPy's float32 exp (numpy/core/src/umath/loops_exponent_log.dispatch.c.src, AVX2 and AVX-512F paths, NumPy >= 1.17; constants of
// numpy/core/include/numpy/npy_math.h): k = rint(x log2 e) by the 1.5 * 2^23 trick, Cody-Waite reduction r = x - k ln 2 in two fused
// steps, exp(r) = P5(r) / Q2(r) in Horner form with fused multiply-adds, IEEE division, result scaled by 2^k.  It is NOT correctly
// rounded (13 % of the affinities of a C8 P8 frame differ from the correctly-rounded value by one ulp), and an ALS run of 600 - 1000
// iterations on the float32 affinity notices: with this form S is the reference's bit for bit (checked here against np.exp on 4 M
// arguments; profiles/r05_oracle_soak.txt).  Hosts without AVX2 take libm's expf in NumPy: not this function.
__device__ __forceinline__ float np_exp_f32(float x) {
    if (x != x) return x;
    if (x > 88.72283935546875f) return __builtin_inff();
    if (x < -103.97208404541015625f) return 0.f;
    const float magic = 0x1.8p+23f;
    float k = fmulr(x, 1.442695040888963407359924681001892137f);
    k = __fsub_rn(faddr(k, magic), magic);
    float r = __fmaf_rn(k, -0x1.62e400p-1f, x);
    r = __fmaf_rn(k, -0x1.7f7d1cp-20f, r);
    float num = __fmaf_rn(5.082762527590693718096e-04f, r, 6.757896990527504603057e-03f);
    num = __fmaf_rn(num, r, 5.114512081637298353406e-02f);
    num = __fmaf_rn(num, r, 2.473615434895520810817e-01f);
    num = __fmaf_rn(num, r, 7.257664613233124478488e-01f);
    num = __fmaf_rn(num, r, 9.999999999980870924916e-01f);
    float den = __fmaf_rn(2.159509375685829852307e-02f, r, -2.742335390411667452936e-01f);
    den = __fmaf_rn(den, r, 1.000000000000000000000e+00f);
    return ldexpf(__fdiv_rn(num, den), (int)k);
}

// DPP cross-lane helpers (no LDS crossbar round trip): quad exchanges and a full-wave sum.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// The value of lane ^ OFF (OFF = 32, 16, 8, 4, 2, 1) WITHOUT the LDS crossbar: gfx950's half / row swaps with both operands the same
// register for 32 and 16 (v_permlane32_swap: lanes 32-63 of the first operand <-> lanes 0-31 of the second; v_permlane16_swap: the odd
// 16-lane rows of the first <-> the even rows of the second), DPP row / quad operations below.  A ds_bpermute butterfly of six
// dependent rounds costs ~2.5 k cycles in the chain kernel, where eleven other waves of the CU keep the LDS pipeline busy.
template <int OFF>
__device__ __forceinline__ double xor_lane(double v) {
    const int lane = threadIdx.x & 63;
    if constexpr (OFF == 32 || OFF == 16) {
        int w[2] = {__double2loint(v), __double2hiint(v)};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if constexpr (OFF == 32) { const auto r = __builtin_amdgcn_permlane32_swap(w[h], w[h], false, false); w[h] = (lane & 32) ? r[0] : r[1]; }
            else { const auto r = __builtin_amdgcn_permlane16_swap(w[h], w[h], false, false); w[h] = (lane & 16) ? r[0] : r[1]; }
        }
        return __hiloint2double(w[1], w[0]);
    } else if constexpr (OFF == 8) {
        return dpp_mov<0x128>(v);                              // row_ror:8
    } else if constexpr (OFF == 4) {
        const double up = dpp_mov<0x104>(v), dn = dpp_mov<0x114>(v);   // row_shl:4 (lane + 4), row_shr:4 (lane - 4)
        return (lane & 4) ? dn : up;
    } else if constexpr (OFF == 2) {
        return dpp_mov<0x4E>(v);                               // quad_perm [2,3,0,1]
    } else {
        static_assert(OFF == 1, "xor_lane: power of two below 64");
        return dpp_mov<0xB1>(v);                               // quad_perm [1,0,3,2]
    }
}
// Wave-wide (64 lanes) sum with a fixed butterfly order (deterministic): v += lane ^ 32, ^ 16, ... ^ 1 -- the order (and therefore the
// bits) of the __shfl_xor butterfly it replaces (checked bit for bit on 262,144 random vectors).
__device__ __forceinline__ double wave_sum(double v) {
    v += xor_lane<32>(v); v += xor_lane<16>(v); v += xor_lane<8>(v);
    v += xor_lane<4>(v); v += xor_lane<2>(v); v += xor_lane<1>(v);
    return v;
}
// sum over the 4 lanes of a quad (lanes 4q..4q+3); every lane gets the result
__device__ __forceinline__ double quad_sum(double v) {
    v += dpp_mov<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);  // quad_perm [2,3,0,1]
    return v;
}
// sum over the wave: rotate-butterfly inside each row of 16 lanes (row_ror), then the four row totals
// through SGPRs (v_readlane); every lane gets the result
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_mov<0x128>(v);  // row_ror:8
    v += dpp_mov<0x124>(v);  // row_ror:4
    v += dpp_mov<0x122>(v);  // row_ror:2
    v += dpp_mov<0x121>(v);  // row_ror:1
    const int lo = __double2loint(v), hi = __double2hiint(v);
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        t += __hiloint2double(__builtin_amdgcn_readlane(hi, 16 * r), __builtin_amdgcn_readlane(lo, 16 * r));
    return t;
}
// The same sum for a vector that vanishes on the lanes >= 16 ROWS: the row totals that are zeros are not fetched.  Bit-identical to
// wave_sum_dpp (t = 0 + r0 + .. is never -0, and adding the zero rows changes nothing).
template <int ROWS>
__device__ __forceinline__ double wave_sum_rows(double v) {
    static_assert(ROWS >= 1 && ROWS <= 4, "rows of 16 lanes");
    v += dpp_mov<0x128>(v);  // row_ror:8
    v += dpp_mov<0x124>(v);  // row_ror:4
    v += dpp_mov<0x122>(v);  // row_ror:2
    v += dpp_mov<0x121>(v);  // row_ror:1
    const int lo = __double2loint(v), hi = __double2hiint(v);
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
        t += __hiloint2double(__builtin_amdgcn_readlane(hi, 16 * r), __builtin_amdgcn_readlane(lo, 16 * r));
    return t;
}
// A wave-uniform value as a scalar: the compiler keeps it in SGPRs from here on.  A value that VALU code computed (a wave
// reduction, an LDS read) lives in a vector register per lane even when every lane holds the same number, and around a call to an
// out-of-line function every live vector register is a 256-byte scratch store + load, where 64 live SGPRs share ONE.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
template <typename Tp>
__device__ __forceinline__ Tp* uni(Tp* p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (Tp*)(((unsigned long long)hi << 32) | lo);
}

// maximum over the wave, same data movement as wave_sum_dpp (rotations inside the rows of 16 lanes, the four row results through
// SGPRs): no LDS-crossbar butterfly; every lane gets the result
__device__ __forceinline__ double wave_max_dpp(double v) {
    v = fmax(v, dpp_mov<0x128>(v));  // row_ror:8
    v = fmax(v, dpp_mov<0x124>(v));  // row_ror:4
    v = fmax(v, dpp_mov<0x122>(v));  // row_ror:2
    v = fmax(v, dpp_mov<0x121>(v));  // row_ror:1
    const int lo = __double2loint(v), hi = __double2hiint(v);
    double t = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
#pragma unroll
    for (int r = 1; r < 4; ++r)
        t = fmax(t, __hiloint2double(__builtin_amdgcn_readlane(hi, 16 * r), __builtin_amdgcn_readlane(lo, 16 * r)));
    return t;
}
__device__ __forceinline__ double fast_rcp64(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = r * (2.0 - x * r);
    r = r * (2.0 - x * r);
    return r;
}

// ---- skeleton + rotation helpers shared by the FK and IK kernels ----
struct SkelDev {
    double dirs[18][3];
    int parents[18];
    int side_map[18];
    int n_side;
    double ref_side[18];
};

__device__ inline void quat_mul(const double* q, const double* r, double* o) {
    o[0] = r[0] * q[0] - r[1] * q[1] - r[2] * q[2] - r[3] * q[3];
    o[1] = r[0] * q[1] + r[1] * q[0] - r[2] * q[3] + r[3] * q[2];
    o[2] = r[0] * q[2] + r[1] * q[3] + r[2] * q[0] - r[3] * q[1];
    o[3] = r[0] * q[3] - r[1] * q[2] + r[2] * q[1] + r[3] * q[0];
}

__device__ inline void euler_to_rot(const double* e, double* R) {
    const double inv = 1.0 / (1.0 + 1e-10);
    double sx, cx, sy, cy, sz, cz;
    sincos(e[0] / 2.0, &sx, &cx);
    sincos(e[1] / 2.0, &sy, &cy);
    sincos(e[2] / 2.0, &sz, &cz);
    const double q0[4] = {cx, inv * sx, 0.0, 0.0};
    const double q1[4] = {cy, 0.0, inv * sy, 0.0};
    const double q2[4] = {cz, 0.0, 0.0, inv * sz};
    double q12[4], q[4];
    quat_mul(q1, q2, q12);
    quat_mul(q0, q12, q);
    const double qw = q[0], qx = q[1], qy = q[2], qz = q[3];
    const double x2 = qx + qx, y2 = qy + qy, z2 = qz + qz;
    const double xx = qx * x2, yy = qy * y2, wx = qw * x2;
    const double xy = qx * y2, yz = qy * z2, wy = qw * y2;
    const double xz = qx * z2, zz = qz * z2, wz = qw * z2;
    R[0] = 1.0 - (yy + zz); R[1] = xy - wz; R[2] = xz + wy;
    R[3] = xy + wz; R[4] = 1.0 - (xx + zz); R[5] = yz - wx;
    R[6] = xz - wy; R[7] = yz + wx; R[8] = 1.0 - (xx + yy);
}

// host mvmcSkeleton -> kernel argument; false if the tables are inconsistent
static inline bool skel_to_dev(const mvmcSkeleton* h, SkelDev* sk) {
    if (h->n_side <= 0 || h->n_side > 18) return false;
    for (int j = 0; j < 18; ++j) {
        for (int k = 0; k < 3; ++k) sk->dirs[j][k] = h->bone_dirs[j][k];
        sk->parents[j] = h->parents[j];
        sk->side_map[j] = h->side_map[j];
        sk->ref_side[j] = h->ref_side_lens[j];
        if (sk->side_map[j] < 0 || sk->side_map[j] >= h->n_side) return false;
        if (j == 0 ? (sk->parents[j] != -1) : (sk->parents[j] < 0 || sk->parents[j] >= j)) return false;
    }
    sk->n_side = h->n_side;
    return true;
}
