// Temporal inverse kinematics, one WAVE per solve (PoseSolver.solve, inverse_kinematics.py:351-433).
//
// SciPy's trf_no_bounds / solve_lsq_trust_region restated with an analytic Jacobian and the normal equations:
// the trust-region step lives in the Krylov tridiagonal basis of (J^T J, g).  A 64-lane wave owns a person-frame:
//   * no workgroup barriers: every reduction is a DPP wave reduction, every exchange an in-wave LDS broadcast;
//   * ~10 KB of LDS (+ the problem's view block) and <= 168 VGPRs per solve: the ~2,500 solves that one time step of the chain
//     protocol launches are all resident at once, and the association kernels of another chain group can share the CU;
//   * the model is built in REDUCED COORDINATES (mvmc_ik_arrow.h: the nine structural null directions of the Euler-angle Jacobian
//     projected out, 30 / 40 columns instead of 39 / 49); J^T J = sum_k D_k^T W_k D_k is accumulated with lane = column, all rows in
//     registers; a lane forms its own column of D_k (cross products of its joints' angular velocities with the lever arms) on the
//     fly, the row-side values come back from LDS as broadcasts, and the tree sparsity of D_k (only ancestors of joint k move it)
//     is a per-joint row mask;
//   * the tridiagonalisation runs on that register image (mvmc_tri_w1.h); the Householder vectors stay in the matrix registers
//     through the first trial and go to a per-solve global scratch only for solves that reject a trial;
//   * trust-region solve, block checks and the cold start are single-wave code (mvmc_eigh_tri.h, mvmc_postopt.h).
// A model that does not split cleanly into range and null space (weakly observed directions, joints no view sees) takes the
// eigensolver fallback in the T basis (tri_eigh_w1).  The Euler-space model (ik1_model_step) remains for a joint at gimbal lock and
// for skeletons the reduced structures are not written for.
#include "mvmc_common.h"
#include "mvmc_postopt.h"
#include "mvmc_eigh_tri.h"
#include "mvmc_tri_w1.h"
#include "mvmc_ik_shared.h"
#include "mvmc_ik_arrow.h"

namespace {

constexpr int NA1 = 50;  // max active parameters

#ifdef MVMC_IK_PROFILE
#define P1_T0 const long long _t0 = clock64();
#ifdef MVMC_TRI_PROFILE
#define P1_ADD(k) if ((threadIdx.x & 63) == 0 && (k) == 2) S.prof[k] += clock64() - _t0;
#else
#define P1_ADD(k) if ((threadIdx.x & 63) == 0) S.prof[k] += clock64() - _t0;
#endif
#else
#define P1_T0
#define P1_ADD(k)
#endif

// phase-local LDS (S.tmp, 256 doubles):
//   FK:       Rl [0,162)  off [162,216)
//   J^T J:    d columns [0,192)
//   tridiag:  vb [0,64)  pb [64,128)
//   checks:   dsc [0,64)  e2 [64,128)  lmul [128,192)  dinv [192,256)
//   tr solve: rh [0,64)  cv [64,128)
// persistent solver vectors (S.sv): D, E, TAU, V0, WN, 64 doubles each
enum { SV_D = 0, SV_E = 64, SV_TAU = 128, SV_V0 = 192, SV_WN = 256, SV_COUNT = 320 };

// skeleton-derived tables: the same for every solve, one copy per workgroup
struct Ik1Tables {
    double dirs[18 * 3], ref_side[MVMC_N_SIDE + 1];   // (the side-length slots: n_side <= MVMC_N_SIDE is checked by every launcher)
    unsigned long long rowmask[2][NOBS];
    int anc[18];
    int smask[18];   // joints (bit j) whose bone length is side-length slot s
    int maxdepth, na[2], n_side;
    signed char depth[18], parents[18], side_map[18];
    signed char lev_list[18], lev_start[20];   // joints ordered by depth; first entry of every level (lev_start[maxdepth + 1] = 18)
    unsigned char act[2][NA1], colkind[2][NA1], cola[2][NA1], colc[2][NA1];
    // reduced coordinates (mvmc_ik_arrow.h).  Reduced column c: rkind 0 = translation along axis rjc; 1 = rotation of joint rja about
    // its Euler axis rjc; 2 = side-length slot rja; 3 = column rjc of the reduced basis of structure rja (the joint pair s_ja, s_jb)
    // (kept small: these tables sit in LDS next to the chain kernel's arena, and 272 bytes more cost it its third workgroup per CU)
    int arrow_ok;                              // the skeleton has the topology the five structures are written for
    unsigned short lenmask[NOBS];              // per observed joint: the length columns (bit i = reduced column 30 + i) on its path
    unsigned char rja[40], rjc[40];
    signed char s_ja[5], s_jb[5], s_tip[5];    // structures: upper joint, lower joint, the lower joint's only child (-1: none)
    signed char e2r[NA1];                      // Euler column (act order of stage 2; stage 1's are its first 39) -> reduced column, or
                                               // -(1 + 8 structure + component)
    // kind of reduced column c: 3 (pair structure) for the limbs and spine + neck, 1 (one joint's rotation) for the head and the root,
    // 0 translation, 2 length
    __host__ __device__ static int rkind(int c) { return c < 16 ? 3 : c < 19 ? 1 : c < 22 ? 0 : c < 25 ? 1 : c < 30 ? 3 : 2; }
    // the reduced columns observed joint k (ancestor mask anc) depends on
    __host__ __device__ unsigned long long rmask(int stage, int k, int anc) const {
        unsigned long long m = 0x3Full << 19;                              // translation and root rotation move every joint
        if ((anc >> 7) & 1 || (anc >> 8) & 1) m |= 0x1Full << 25;          // spine + neck
        if (stage) m |= (unsigned long long)lenmask[k] << 30;
        if ((anc >> 1) & 1) m |= 0xFull;                                   // the limb (or the head) below which it hangs
        if ((anc >> 4) & 1) m |= 0xFull << 4;
        if ((anc >> 9) & 1) m |= 0xFull << 8;
        if ((anc >> 12) & 1) m |= 0xFull << 12;
        if ((anc >> 15) & 1) m |= 0x7ull << 16;
        return m;
    }
};

// per-solve state (one wave).  The problem's observations are NOT part of it -- and since round 5 not in LDS at all: the evaluation
// reads a view's keypoint rows (COCO-17; the synthetic mid-spine row, inverse_kinematics.py:339-348, is formed from its four rows on the
// fly) and its projection matrix straight from the read-only input tensors (15 doubles per lane and view, L1 / L2 hits after the first
// touch: a frame's keypoints are 8 KB), through a MEMBER LIST of (pose index, camera) per view that the caller hands to ik1_solve: the
// stand-alone kernel a list of v_max entries per workgroup, the chain kernel a slice of ONE list per workgroup -- the clusters of a frame
// are disjoint sets of the frame's poses, so the lists of all of a frame's problems, laid end to end, never exceed the number of poses
// in the frame, whatever the size of a single cluster (the reference has no cap on it: motion_capture.py:417-446, :618-626).  The
// 12.7 KB view pool (24 views x 66 doubles) this replaces was what kept the SMALL layout's arena above 40,960 B, i.e. a fourth
// workgroup per CU out of reach.
struct Ik1Shared {
    __attribute__((aligned(16))) double tmp[256];
    double sv[SV_COUNT];
    double x[68], xn[68], side[MVMC_N_SIDE + 1];
    double Rg[18 * 9], pos[18 * 3], bvec[18 * 3];
    double hs[18 * 4];      // sin, cos of half the x and y Euler angles of every joint (from the last FK)
    double Wk[NOBS * 6], tk[NOBS * 3];
    double sc[12];          // {|g|^2, |g|_inf, alpha, pred, beta0, tau0, |J^T J|_1, coupling, |step|, |x|, fallback rows}
    int nviews, mode3d;
    // the member list (int32 pose index / uint16 camera per view), in units of TWO bytes from the start of this struct (the lists lie
    // behind the solve blocks of their arena: up to 78 KB away in the BIG layout)
    unsigned short mq_off2, mc_off2;
    // pairing word (mvmc_ik_pair.h): the even wave of a pair holds the pair's mailbox here, the odd wave its completion counter
    int pairw;
#ifdef MVMC_IK_PROFILE
    long long prof[8];
#endif
};
constexpr int MVMC_IK_VIEW_DOUBLES = 54 + 12;   // a staged view (cold start only, in the solve's global scratch): 18 rows + P
// where a solve's observations come from: the read-only inputs (reprojection mode) or the 3-D targets (mode3d); wave-uniform
struct Ik1Obs {
    const mvmc_gdouble* kps17;   // (.., 17, 3)
    const mvmc_gdouble* Pm;      // (C, 3, 4)
    const mvmc_gdouble* tg;      // this problem's 18 x {x, y, z, weight} targets, or NULL
};
// (derived from &S, which every out-of-line function declares to be LDS: the accesses stay ds_ instructions)
__device__ __forceinline__ const int* ik1_mq(Ik1Shared& S) { return reinterpret_cast<const int*>(reinterpret_cast<const char*>(&S) + 2 * (int)S.mq_off2); }
__device__ __forceinline__ const unsigned short* ik1_mc(Ik1Shared& S) { return reinterpret_cast<const unsigned short*>(reinterpret_cast<const char*>(&S) + 2 * (int)S.mc_off2); }

__device__ __forceinline__ double wave_max64(double v) { return wave_max_dpp(v); }

// R = Rx Ry Rz through the reference's quaternion product (common.h: euler_to_rot), also returning the half-angle
// sines / cosines of the first two angles
__device__ inline void rot_from_half_angles(double sx, double cx, double sy, double cy, double sz, double cz, double* R) {
    const double inv = 1.0 / (1.0 + 1e-10);
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

// ---------------------------------------------------------------------------------------------
// FK + residual; with want_jac also the per-joint normal-equation blocks W_k (S.Wk) and t_k (S.tk).
// Lane (k, r) = (lane & 15, lane >> 4) handles observed joint k in the views r, r + 4.  Returns 0.5 |f|^2.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double ik1_eval(Ik1Shared& S, const Ik1Tables& T, const double* xs, int stage, bool want_jac, const Ik1Obs& O) {
    const int lane = threadIdx.x & 63;
    double* Rl = S.tmp;
    double* off = S.tmp + 162;
    // the 54 half-angle sincos on 54 lanes (they are the long pole of this phase: three in a row on 18 lanes before), then joint lanes
    double* szc = S.tmp + 216;   // sin, cos of z / 2 per joint (36 doubles behind Rl and off)
    if (lane < 54) {
        const int j = lane / 3, a = lane - 3 * j;
        double sn, cs;
        sincos(xs[3 + lane] / 2.0, &sn, &cs);
        if (a < 2) { S.hs[j * 4 + 2 * a] = sn; S.hs[j * 4 + 2 * a + 1] = cs; }
        else { szc[j * 2] = sn; szc[j * 2 + 1] = cs; }
    }
    MVMC_WAVE_SYNC();
    if (lane < 18) {
        const double* h = &S.hs[lane * 4];
        rot_from_half_angles(h[0], h[1], h[2], h[3], szc[lane * 2], szc[lane * 2 + 1], &Rl[lane * 9]);
        double len = 0.0;
        if (lane > 0) len = (stage == 0) ? S.side[T.side_map[lane]] : xs[57 + T.side_map[lane]];
        for (int k = 0; k < 3; ++k) off[lane * 3 + k] = T.dirs[lane * 3 + k] * len;
    }
    MVMC_WAVE_SYNC();
    if (lane < 9) S.Rg[lane] = Rl[lane];
    if (lane < 3) S.pos[lane] = xs[lane];
    MVMC_WAVE_SYNC();
    for (int lev = 1; lev <= T.maxdepth; ++lev) {
        const int l0 = T.lev_start[lev], l1 = T.lev_start[lev + 1];      // the level's joints: no pass over the other joints' entries
        for (int t = lane; t < (l1 - l0) * 9; t += 64) {
            const int j = T.lev_list[l0 + t / 9], e = t - (t / 9) * 9;
            {
                const int p = T.parents[j], r = e / 3, c = e - r * 3;
                const double* Gp = &S.Rg[p * 9];
                const double* Rj = &Rl[j * 9];
                S.Rg[j * 9 + e] = Gp[r * 3] * Rj[c] + Gp[r * 3 + 1] * Rj[3 + c] + Gp[r * 3 + 2] * Rj[6 + c];
                if (e < 3) {
                    S.pos[j * 3 + e] = Gp[e * 3] * off[j * 3] + Gp[e * 3 + 1] * off[j * 3 + 1] +
                                       Gp[e * 3 + 2] * off[j * 3 + 2] + S.pos[p * 3 + e];
                    S.bvec[j * 3 + e] = Gp[e * 3] * T.dirs[j * 3] + Gp[e * 3 + 1] * T.dirs[j * 3 + 1] +
                                        Gp[e * 3 + 2] * T.dirs[j * 3 + 2];
                }
            }
        }
        MVMC_WAVE_SYNC();
    }
    const int k = lane & 15, r = lane >> 4;
    const int nviews = uni(S.nviews);
    const int* mq = ik1_mq(S);
    const unsigned short* mc = ik1_mc(S);
    const double* X = &S.pos[kIkSkel[k] * 3];
    const double X0 = X[0], X1 = X[1], X2 = X[2];
    const int obs = kIkObs[k];
    double f2 = 0.0, o[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (S.mode3d) {
        // residual (pos_k - target_k) * w_k (inverse_kinematics.py:280-336): J_k = w_k D_k, so W_k = w_k^2 I, t_k = w_k f_k
        if (r == 0) {
            const mvmc_gdouble* tg = O.tg + obs * 4;
            const double w = tg[3], w2 = w * w;
            const double f0 = (X0 - tg[0]) * w, f1 = (X1 - tg[1]) * w, f2c = (X2 - tg[2]) * w;
            f2 = f0 * f0 + f1 * f1 + f2c * f2c;
            o[0] = w2; o[3] = w2; o[5] = w2;
            o[6] = w * f0; o[7] = w * f1; o[8] = w * f2c;
        }
    } else
    for (int v = r; v < nviews; v += 4) {
        const mvmc_gdouble* kp = O.kps17 + (size_t)mq[v] * 51;
        const mvmc_gdouble* P = O.Pm + (int)mc[v] * 12;
        // the observation row: a COCO-17 keypoint, or the mid-spine (a quarter of shoulders + hips, the product of their scores)
        double ob0, ob1, s;
        if (obs < 17) { ob0 = kp[obs * 3]; ob1 = kp[obs * 3 + 1]; s = kp[obs * 3 + 2]; }
        else {
            const double sh0 = 0.5 * (kp[5 * 3] + kp[6 * 3]), hp0 = 0.5 * (kp[11 * 3] + kp[12 * 3]);
            const double sh1 = 0.5 * (kp[5 * 3 + 1] + kp[6 * 3 + 1]), hp1 = 0.5 * (kp[11 * 3 + 1] + kp[12 * 3 + 1]);
            ob0 = 0.5 * (sh0 + hp0); ob1 = 0.5 * (sh1 + hp1);
            s = kp[5 * 3 + 2] * kp[6 * 3 + 2];
            s *= kp[11 * 3 + 2] * kp[12 * 3 + 2];
        }
        const double P0 = P[0], P1 = P[1], P2 = P[2], P3 = P[3], P4 = P[4], P5 = P[5], P6 = P[6], P7 = P[7], P8 = P[8], P9 = P[9],
                     P10 = P[10], P11 = P[11];
        const double h0 = P0 * X0 + P1 * X1 + P2 * X2 + P3;
        const double h1 = P4 * X0 + P5 * X1 + P6 * X2 + P7;
        const double h2 = P8 * X0 + P9 * X1 + P10 * X2 + P11;
        const double w = 1e-5 + h2, iw = 1.0 / w;
        const double u = h0 / w, vv = h1 / w;
        const double fu = (u - ob0) * s, fv = (vv - ob1) * s;
        f2 += fu * fu + fv * fv;
        if (want_jac) {
            const double Pr0[3] = {P0, P1, P2}, Pr1[3] = {P4, P5, P6}, Pr2[3] = {P8, P9, P10};
            double du[3], dv[3];
            for (int c = 0; c < 3; ++c) {
                du[c] = (Pr0[c] - u * Pr2[c]) * iw;
                dv[c] = (Pr1[c] - vv * Pr2[c]) * iw;
            }
            const double s2 = s * s;
            o[0] += s2 * (du[0] * du[0] + dv[0] * dv[0]);
            o[1] += s2 * (du[0] * du[1] + dv[0] * dv[1]);
            o[2] += s2 * (du[0] * du[2] + dv[0] * dv[2]);
            o[3] += s2 * (du[1] * du[1] + dv[1] * dv[1]);
            o[4] += s2 * (du[1] * du[2] + dv[1] * dv[2]);
            o[5] += s2 * (du[2] * du[2] + dv[2] * dv[2]);
            o[6] += s * (du[0] * fu + dv[0] * fv);
            o[7] += s * (du[1] * fu + dv[1] * fv);
            o[8] += s * (du[2] * fu + dv[2] * fv);
        }
    }
    const double cost = 0.5 * wave_sum(f2);
    if (want_jac) {
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            o[e] += xor_lane<16>(o[e]);
            o[e] += xor_lane<32>(o[e]);
        }
        if (lane < NOBS) {
#pragma unroll
            for (int e = 0; e < 6; ++e) S.Wk[lane * 6 + e] = o[e];
#pragma unroll
            for (int e = 0; e < 3; ++e) S.tk[lane * 3 + e] = o[6 + e];
        }
    }
    MVMC_WAVE_SYNC();
    return cost;
}

// ---------------------------------------------------------------------------------------------
// The solver is a small driver (ik1_trf: wave-uniform control flow, its state in scalar registers) around three
// out-of-line functions, so that NO vector register is live across a call -- a live one costs a 256-byte scratch store
// and a load per call, and the model function needs every register the occupancy allows:
//   ik1_eval_nl         FK + residual (+ blocks) at S.x or S.xn
//   ik1_model_step      the trust-region model at the last evaluated point AND the first trial step from it
//   ik1_fallback_trial  a trial step of a model that took the eigenbasis fallback
//   ik1_retry_trial     a further trial step of a common-path model whose reflectors are in the global scratch
// (Tried again on this structure: everything inlined into one state machine with a single evaluation site and a single model site per
// row count -- no calls, no per-call save of the register that holds spilled SGPRs.  The allocator then keeps ~40 loop-invariant values
// of the driver and the evaluation in spill slots and reloads them inside the Householder loop: 192 scratch stores / 1,140 loads in the
// IK phase.  The out-of-line boundaries are what keeps the model's register allocation to itself.)
// Results of a trial: S.xn = the trial point, S.sc[2] = alpha, S.sc[3] = predicted reduction, S.sc[8] = |step|,
// S.sc[9] = |x|.
// ---------------------------------------------------------------------------------------------
__device__ __noinline__ double ik1_eval_nl(Ik1Shared& S, const Ik1Tables& T, int at_trial, int stage, bool want_jac,
                                           const mvmc_gdouble* kps17, const mvmc_gdouble* Pm, const mvmc_gdouble* tg) {
    MVMC_ASSUME_LDS(&S);
    MVMC_ASSUME_LDS(&T);
    const Ik1Obs O = {uni(kps17), uni(Pm), uni(tg)};   // (arrive in vector registers; wave-uniform: scalar bases of the loads)
    return ik1_eval(S, T, at_trial ? S.xn : S.x, stage, want_jac, O);
}

// S.xn = S.x + step on the active parameters (stepj: lane j's component), S.sc[9] = |x|
__device__ __forceinline__ void ik1_trial_point(Ik1Shared& S, const Ik1Tables& T, int stage, int na, double stepj) {
    const int lane = threadIdx.x & 63;
    const int nfull = (stage == 0) ? 57 : 57 + T.n_side;
    double xx = (lane < nfull) ? S.x[lane] * S.x[lane] : 0.0;
    if (lane + 64 < nfull) xx += S.x[lane + 64] * S.x[lane + 64];
    const double x_norm = sqrt(wave_sum(xx));
    if (lane == 0) S.sc[9] = x_norm;
    if (lane < nfull) S.xn[lane] = S.x[lane];
    if (lane + 64 < nfull) S.xn[lane + 64] = S.x[lane + 64];
    MVMC_WAVE_SYNC();
    if (lane < na) { const int f = T.act[stage][lane]; S.xn[f] = S.x[f] + stepj; }
    MVMC_WAVE_SYNC();
}

// ---------------------------------------------------------------------------------------------
// Trust-region model at the last evaluated point (FK state, Wk, tk, hs in LDS):
//   g = D^T t (one component per lane),  J^T J = sum_k D_k^T W_k D_k into registers,
//   then -- unless the gradient test or the evaluation budget stops the iteration (the caller's next step is a
//   break) -- the Krylov tridiagonalisation, the leading-block checks and, on the common path, the trial step for
//   (Delta, alpha): the trust-region solve in the tridiagonal basis and Q applied from the reflectors that are
//   still in the matrix registers.  The first trial a solve rejects is re-made by calling the function again (after
//   re-evaluating S.x) with the new radius and dump = true: the model is a deterministic function of the LDS state, 17 %
//   of the warm solves pay that second build once, and the others never send a Householder vector to memory.
// *mode_out = 0 (stopped after g), 1 (trial made), 2 (eigenbasis path on m = S.sc[10] rows of T: lam in the WN slot, suf in
// the D slot, vectors and reflectors in the global scratch; the trial is ik1_fallback_trial's).
// S.sc[0] = |g|^2, S.sc[1] = |g|_inf, S.sc[4..8) = {beta0, tau0, |J^T J|_1, coupling}.
// ---------------------------------------------------------------------------------------------
// DBG (mvmc_debug_ik_model_step only; the solver's instances are DBG = false): the gradient also goes to hh[MVMC_IK_DBG_G + lane].
constexpr int MVMC_IK_DBG_G = 6500;
template <int N, bool DBG = false>
__device__ __noinline__ void ik1_model_step(Ik1Shared& S, const Ik1Tables& T, int stage, bool budget_left, double gtol,
                                            mvmc_gdouble* __restrict__ hh, double Delta, double alpha, bool dump, int* mode_out) {
    // S and T arrive as generic pointers (the function is out of line); telling the compiler that they are LDS turns every
    // access below into a ds_ instruction instead of a flat_ one (InferAddressSpaces uses the assumption)
    MVMC_ASSUME_LDS(&S);
    MVMC_ASSUME_LDS(&T);
    hh = uni(hh);   // (arrives in vector registers; as a scalar base it costs no register next to the matrix rows)
    const int lane = threadIdx.x & 63;
    const int na = uni(T.na[stage]);
    const bool on = lane < na;
    const int cl = on ? lane : 0;
    const int kind = T.colkind[stage][cl], ja = T.cola[stage][cl], jc = T.colc[stage][cl];
    // own rotation axis in the world frame and own pivot (kind 1): R_a = Rx Ry Rz inside the parent's frame
    double ax0 = 0.0, ax1 = 0.0, ax2 = 0.0, pa0 = 0.0, pa1 = 0.0, pa2 = 0.0;
#ifdef MVMC_IK_PROFILE
    long long _tp = clock64();
#ifdef MVMC_TRI_PROFILE
#define M1STAMP(k) { const long long _t = clock64(); if (lane == 0 && (k) == 5) S.prof[k] += _t - _tp; _tp = _t; }
#else
#define M1STAMP(k) { const long long _t = clock64(); if (lane == 0) S.prof[k] += _t - _tp; _tp = _t; }
#endif
#else
#define M1STAMP(k)
#endif
    if (kind == 1) {
        const double* h = &S.hs[ja * 4];
        const double s0 = 2.0 * h[0] * h[1], c0 = h[1] * h[1] - h[0] * h[0];
        const double s1 = 2.0 * h[2] * h[3], c1 = h[3] * h[3] - h[2] * h[2];
        double l0, l1, l2;
        if (jc == 0) { l0 = 1.0; l1 = 0.0; l2 = 0.0; }
        else if (jc == 1) { l0 = 0.0; l1 = c0; l2 = s0; }
        else { l0 = s1; l1 = -s0 * c1; l2 = c0 * c1; }
        if (ja == 0) { ax0 = l0; ax1 = l1; ax2 = l2; }
        else {
            const double* Gp = &S.Rg[T.parents[ja] * 9];
            ax0 = Gp[0] * l0 + Gp[1] * l1 + Gp[2] * l2;
            ax1 = Gp[3] * l0 + Gp[4] * l1 + Gp[5] * l2;
            ax2 = Gp[6] * l0 + Gp[7] * l1 + Gp[8] * l2;
        }
        pa0 = S.pos[ja * 3]; pa1 = S.pos[ja * 3 + 1]; pa2 = S.pos[ja * 3 + 2];
    }
    // the lane's axis and pivot wait in LDS (the solver vectors are dead while the model is rebuilt) and are re-read per
    // joint: six doubles less to keep in registers next to the N matrix rows
    double* axl = S.sv;
    axl[lane] = ax0; axl[64 + lane] = ax1; axl[128 + lane] = ax2;
    axl[192 + lane] = pa0; axl[256 + lane] = pa1;
    S.tmp[192 + lane] = pa2;
    MVMC_WAVE_SYNC();
    double a[N];
#pragma unroll
    for (int i = 0; i < N; ++i) a[i] = 0.0;
    double gj = 0.0;
    double* db = S.tmp;
    for (int k = 0; k < NOBS; ++k) {
        const int K = kIkSkel[k];
        double d0 = 0.0, d1 = 0.0, d2 = 0.0;
        if (on) {
            if (kind == 0) {
                d0 = jc == 0 ? 1.0 : 0.0; d1 = jc == 1 ? 1.0 : 0.0; d2 = jc == 2 ? 1.0 : 0.0;
            } else if (kind == 1) {
                if ((T.anc[K] >> ja) & 1) {
                    const double r0 = S.pos[K * 3] - axl[192 + lane], r1 = S.pos[K * 3 + 1] - axl[256 + lane],
                                 r2 = S.pos[K * 3 + 2] - S.tmp[192 + lane];
                    const double x0 = axl[lane], x1 = axl[64 + lane], x2 = axl[128 + lane];
                    d0 = x1 * r2 - x2 * r1; d1 = x2 * r0 - x0 * r2; d2 = x0 * r1 - x1 * r0;
                }
            } else {
                // bones on the path root .. K whose length is slot ja, from K upwards (parents have smaller indices: descending j) -- the
                // set is a mask intersection, so the lane does not walk the tree through dependent LDS reads
                unsigned path = ((unsigned)T.anc[K] | (1u << K)) & (unsigned)T.smask[ja] & ~1u;
                while (path) {
                    const int j = 31 - __builtin_clz(path);
                    path &= ~(1u << j);
                    d0 += S.bvec[j * 3]; d1 += S.bvec[j * 3 + 1]; d2 += S.bvec[j * 3 + 2];
                }
            }
        }
        const double* W = &S.Wk[k * 6];
        const double y0 = W[0] * d0 + W[1] * d1 + W[2] * d2;
        const double y1 = W[1] * d0 + W[3] * d1 + W[4] * d2;
        const double y2 = W[2] * d0 + W[4] * d1 + W[5] * d2;
        gj += d0 * S.tk[k * 3] + d1 * S.tk[k * 3 + 1] + d2 * S.tk[k * 3 + 2];
        MVMC_WAVE_SYNC();  // the previous joint's broadcasts are done
        db[lane * 3] = d0; db[lane * 3 + 1] = d1; db[lane * 3 + 2] = d2;
        MVMC_WAVE_SYNC();
        // rows in chunks behind one wave-uniform test each (d_i vanishes on the other rows of a live chunk): the chunk's
        // broadcast ds_read_b128 are in flight together instead of one LDS round trip per row.  Four rows per chunk: the 50-row
        // instance has no registers for more operands, and in the 40-row one 8 rows per chunk measured slower (fewer chunks are
        // skipped by the tree sparsity than round trips are saved: IK 37.6 -> 38.6 M cycles per chain)
#ifdef MVMC_IK_GR
        constexpr int GR = MVMC_IK_GR, GL = GR * 3 / 2;
#else
        constexpr int GR = 4, GL = GR * 3 / 2;
#endif
        const unsigned long long m = T.rowmask[stage][k];
        const unsigned mlo = __builtin_amdgcn_readfirstlane((unsigned)m), mhi = __builtin_amdgcn_readfirstlane((unsigned)(m >> 32));
#pragma unroll
        for (int c = 0; c < N; c += GR) {
            const unsigned bits = (c < 32 ? (mlo >> (c & 31)) : (mhi >> (c & 31))) & ((1u << GR) - 1u);
            if (bits) {
                double2 t[GL];
#pragma unroll
                for (int u = 0; u < GL; ++u) t[u] = *reinterpret_cast<const double2*>(&db[c * 3 + 2 * u]);
                const double* tt = reinterpret_cast<const double*>(t);
#pragma unroll
                for (int i = 0; i < GR; ++i)
                    if (c + i < N) a[c + i] += tt[3 * i] * y0 + tt[3 * i + 1] * y1 + tt[3 * i + 2] * y2;
            }
        }
    }
    MVMC_WAVE_SYNC();
    const double gg = uni(wave_sum_dpp(gj * gj)), ginf = uni(wave_max64(fabs(gj)));
    if (lane == 0) { S.sc[0] = gg; S.sc[1] = ginf; }
    if (DBG) hh[MVMC_IK_DBG_G + lane] = on ? gj : 0.0;
    M1STAMP(4)
    if (ginf < gtol || !budget_left) { *mode_out = 0; return; }
    double scv, tauv;
    int ksteps;
    const int kk = uni(eightri::tridiag_krylov_w1<N>(a, gj, na, S.sv + SV_D, S.sv + SV_E, S.sv + SV_TAU, S.sv + SV_V0, S.tmp, S.tmp + 64,
                                                     &S.sc[4], scv, tauv, ksteps
#ifdef MVMC_TRI_PROFILE
                                                     , S.prof
#endif
                                                     ));
    // the reflectors, two rows to a register: the matrix registers die here, before the checks and the trust-region solve
    double pk[(N - 2) / 2];
    eightri::pack_reflectors<N>(a, pk);
    if (dump) eightri::dump_reflectors<N>(pk, scv, ksteps, na, hh);   // a solve that has rejected a trial before: see ik1_trf
    M1STAMP(5)
    bool ok = kk > 0;
    if (ok) ok = eightri::krylov_block_ok(S.sv + SV_D, S.sv + SV_E, kk, na, S.sc[6], S.sc[7], S.tmp, S.tmp + 64, S.tmp + 128,
                                          S.tmp + 192, S.sv + SV_WN);
    MVMC_WAVE_SYNC();
    M1STAMP(6)
    if (ok) {
        // ---- the trial step, in the tridiagonal basis ----
        const double beta0 = S.sc[4], tau0 = S.sc[5], pivmin = 1e-16 * S.sc[6] + 1e-300, coupling = S.sc[7];
        double* rh = S.tmp;
        double* cv = S.tmp + 64;
        rh[lane] = lane == 0 ? beta0 : 0.0;
        MVMC_WAVE_SYNC();
        double pred, step_norm;
        alpha = eightri::tr_solve_tri<false, N>(S.sv + SV_D, S.sv + SV_E, rh, kk, Delta, alpha, gg, pivmin, nullptr, nullptr, nullptr,
                                                nullptr, cv, &pred, &step_norm);
        MVMC_WAVE_SYNC();
        double c = lane < kk ? cv[lane] : 0.0;
        if (kk < na) {
            // component along the first null coordinate: keeps the step orthogonal to the null vector
            const double eta = coupling * wave_sum_dpp(lane < kk ? S.sv[SV_WN + lane] * c : 0.0);
            if (lane == kk) c = eta;
        }
        const double stepj = eightri::apply_q_packed<N>(pk, scv, tauv, S.sv + SV_V0, tau0, kk, na, c);
        if (lane == 0) { S.sc[2] = alpha; S.sc[3] = pred; S.sc[8] = step_norm; S.sc[11] = (double)kk; }
        ik1_trial_point(S, T, stage, na, stepj);
        M1STAMP(3)
        *mode_out = 1;
        return;
    }
    // No clean split between range and null space (weakly observed directions, missing joints): the step is taken
    // in the eigenbasis of the tridiagonal matrix instead, with the numerically-null cluster removed -- the
    // eigensolver fallback in the T basis, where suf = V^T g = beta0 * (first components).
    //   unclean collapse:  T is complete (na rows);   clean collapse: the leading block plus its coupling row.
    // Its trials outlive this function: the reflectors go to the global scratch.
    if (!dump) eightri::dump_reflectors<N>(pk, scv, ksteps, na, hh);
    const int m = kk < 0 ? na : (kk < na ? kk + 1 : na);
    mvmc_gdouble* Zg = hh + 64 * NA1;
    eightri::tri_eigh_w1<N>(S.sv + SV_D, S.sv + SV_E, m, S.sv + SV_WN, Zg, S.tmp, S.tmp + 64, S.tmp + 128);
    const double suf = lane < m ? S.sc[4] * Zg[lane] : 0.0;
    MVMC_WAVE_SYNC();
    S.sv[SV_D + lane] = suf;   // d, e are dead: lam lives in the WN slot, suf in the D slot
    if (lane == 0) S.sc[10] = (double)m;
    MVMC_WAVE_SYNC();
    *mode_out = 2;
}

// ---------------------------------------------------------------------------------------------
// The same model in REDUCED COORDINATES (mvmc_ik_arrow.h): the nine structural null directions of the Euler-angle Jacobian are
// projected out by an orthonormal change of basis per limb, so the matrix has 30 / 40 rows and columns instead of 39 / 49 -- every
// Householder step and every row of the accumulation is a quarter shorter -- and the Krylov space no longer ends in a structural
// null block.  Everything else (tridiagonalisation from the gradient, block checks, trust-region solve, eigenbasis fallback) is the
// code of ik1_model_step on the smaller matrix; the step is mapped back to Euler space before the trial point is formed.
// *mode_out as ik1_model_step, plus 3 = a joint at gimbal lock (the closed-form null vectors need cos(e_y) != 0): not applicable.
// ---------------------------------------------------------------------------------------------
// (Round 6, measured: the LAST model of a truncated stage -- no evaluation left: 2 of a warm solve's 10 models -- is only asked for its
// gradient, the test SciPy makes before it looks at the budget (trf.py:451-460).  An instance of this function without the matrix for
// those calls, bit-identical and free of scratch: 550 k -> 535 k frames/s on the chain kernel, twice, with the IK phase 6 % LONGER per
// chain.  Not kept; docs/design_measurement.md, "Round 6".)
template <int N, int STAGE, bool DBG = false>
__device__ __noinline__ void ik1_model_step_r(Ik1Shared& S, const Ik1Tables& T, bool budget_left, double gtol,
                                            mvmc_gdouble* __restrict__ hh, double Delta, double alpha, bool dump, int* mode_out) {
    // S and T arrive as generic pointers (the function is out of line); telling the compiler that they are LDS turns every
    // access below into a ds_ instruction instead of a flat_ one (InferAddressSpaces uses the assumption)
    MVMC_ASSUME_LDS(&S);
    MVMC_ASSUME_LDS(&T);
    hh = uni(hh);   // (arrives in vector registers; as a scalar base it costs no register next to the matrix rows)
    const int lane = threadIdx.x & 63;
    using namespace arrow;
    constexpr int stage = STAGE, NR = Dim<STAGE>::NR;
    static_assert(N == NR, "one matrix row per reduced column");
    const int na = NR;                        // the model's dimension: reduced columns (mvmc_ik_arrow.h)
    const int nae = uni(T.na[STAGE]);         // Euler-space columns (the trial point)
    const bool on = lane < na;
    const int kind = on ? Ik1Tables::rkind(lane) : 4;
#ifdef MVMC_IK_PROFILE
    long long _tp = clock64();
#ifdef MVMC_TRI_PROFILE
#define M1STAMP(k) { const long long _t = clock64(); if (lane == 0 && (k) == 5) S.prof[k] += _t - _tp; _tp = _t; }
#else
#define M1STAMP(k) { const long long _t = clock64(); if (lane == 0) S.prof[k] += _t - _tp; _tp = _t; }
#endif
#else
#define M1STAMP(k)
#endif
    // ---- the reflectors of the five structures (lanes 0 .. 4), from the closed-form null vectors; they wait in S.xn, which nothing
    // touches until the trial point is formed at the very end ----
    double* refl = S.xn;
    bool bad = false;
    if (lane < 5) bad = structure_reflectors(S.pos, S.hs, S.Rg, T, lane, refl + lane * 11);
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull) { *mode_out = 3; return; }   // gimbal lock: the caller takes the Euler-space model
    MVMC_WAVE_SYNC();
    // ---- the lane's column: angular velocities of its (up to two) joints, in the world frame ----
    int ja = 0, jb = 0;
    double wa0 = 0.0, wa1 = 0.0, wa2 = 0.0, wb0 = 0.0, wb1 = 0.0, wb2 = 0.0;
    if (kind == 1) {
        ja = jb = T.rja[lane];
        const int jc = T.rjc[lane];
        const double b[3] = {jc == 0 ? 1.0 : 0.0, jc == 1 ? 1.0 : 0.0, jc == 2 ? 1.0 : 0.0};
        double w[3];
        omega_of(&S.hs[ja * 4], &S.Rg[(ja ? T.parents[ja] : 0) * 9], ja == 0, b, w);
        wa0 = w[0]; wa1 = w[1]; wa2 = w[2];
    } else if (kind == 3) {
        const int s = T.rja[lane], bc = T.rjc[lane];
        ja = T.s_ja[s]; jb = T.s_jb[s];
        double z[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) z[i] = bc == i ? 1.0 : 0.0;
        apply_reflectors(refl + s * 11, z);
        double w[3];
        omega_of(&S.hs[ja * 4], &S.Rg[T.parents[ja] * 9], false, z, w);
        wa0 = w[0]; wa1 = w[1]; wa2 = w[2];
        omega_of(&S.hs[jb * 4], &S.Rg[T.parents[jb] * 9], false, z + 3, w);
        wb0 = w[0]; wb1 = w[1]; wb2 = w[2];
    }
    const int jl = (kind == 2) ? (int)T.rja[lane] : 0;     // length slot
    const int jcc = (kind == 0) ? (int)T.rjc[lane] : 0;
#ifdef MVMC_IK_PARK
    // (128-register build: the lane's two angular velocities wait in LDS -- the solver vectors are dead while the model is built -- and
    // are re-read per joint: twelve registers less to keep next to the N matrix rows, where they would otherwise travel through scratch)
    double* wl = S.sv;
    wl[lane] = wa0; wl[64 + lane] = wa1; wl[128 + lane] = wa2; wl[192 + lane] = wb0; wl[256 + lane] = wb1; S.tmp[192 + lane] = wb2;
    MVMC_WAVE_SYNC();
#endif
    double a[N];
#pragma unroll
    for (int i = 0; i < N; ++i) a[i] = 0.0;
    double gj = 0.0;
    double* db = S.tmp;
    for (int k = 0; k < NOBS; ++k) {
        const int K = kIkSkel[k];
        double d0 = 0.0, d1 = 0.0, d2 = 0.0;
        if (kind == 0) {
            d0 = jcc == 0 ? 1.0 : 0.0; d1 = jcc == 1 ? 1.0 : 0.0; d2 = jcc == 2 ? 1.0 : 0.0;
        } else if (kind == 1 || kind == 3) {
            const int anc = T.anc[K];
#ifdef MVMC_IK_PARK
            const double wa0 = wl[lane], wa1 = wl[64 + lane], wa2 = wl[128 + lane];
#endif
            if ((anc >> ja) & 1) {
                const double r0 = S.pos[K * 3] - S.pos[ja * 3], r1 = S.pos[K * 3 + 1] - S.pos[ja * 3 + 1], r2 = S.pos[K * 3 + 2] - S.pos[ja * 3 + 2];
                d0 = wa1 * r2 - wa2 * r1; d1 = wa2 * r0 - wa0 * r2; d2 = wa0 * r1 - wa1 * r0;
            }
            if (kind == 3 && ((anc >> jb) & 1)) {
#ifdef MVMC_IK_PARK
                const double wb0 = wl[192 + lane], wb1 = wl[256 + lane], wb2 = S.tmp[192 + lane];
#endif
                const double r0 = S.pos[K * 3] - S.pos[jb * 3], r1 = S.pos[K * 3 + 1] - S.pos[jb * 3 + 1], r2 = S.pos[K * 3 + 2] - S.pos[jb * 3 + 2];
                d0 += wb1 * r2 - wb2 * r1; d1 += wb2 * r0 - wb0 * r2; d2 += wb0 * r1 - wb1 * r0;
            }
        } else if (kind == 2) {
            unsigned path = ((unsigned)T.anc[K] | (1u << K)) & (unsigned)T.smask[jl] & ~1u;
            while (path) {
                const int j = 31 - __builtin_clz(path);
                path &= ~(1u << j);
                d0 += S.bvec[j * 3]; d1 += S.bvec[j * 3 + 1]; d2 += S.bvec[j * 3 + 2];
            }
        }
        const double* W = &S.Wk[k * 6];
        const double y0 = W[0] * d0 + W[1] * d1 + W[2] * d2;
        const double y1 = W[1] * d0 + W[3] * d1 + W[4] * d2;
        const double y2 = W[2] * d0 + W[4] * d1 + W[5] * d2;
        gj += d0 * S.tk[k * 3] + d1 * S.tk[k * 3 + 1] + d2 * S.tk[k * 3 + 2];
        MVMC_WAVE_SYNC();  // the previous joint's broadcasts are done
        db[lane * 3] = d0; db[lane * 3 + 1] = d1; db[lane * 3 + 2] = d2;
        MVMC_WAVE_SYNC();
        // rows in chunks behind one wave-uniform test each (d_i vanishes on the other rows of a live chunk): the chunk's
        // broadcast ds_read_b128 are in flight together instead of one LDS round trip per row.  Four rows per chunk: the 50-row
        // instance has no registers for more operands, and in the 40-row one 8 rows per chunk measured slower (fewer chunks are
        // skipped by the tree sparsity than round trips are saved: IK 37.6 -> 38.6 M cycles per chain)
#ifdef MVMC_IK_GR
        constexpr int GR = MVMC_IK_GR, GL = GR * 3 / 2;
#else
        constexpr int GR = 4, GL = GR * 3 / 2;
#endif
        const unsigned long long m = T.rmask(STAGE, k, T.anc[K]);
        const unsigned mlo = __builtin_amdgcn_readfirstlane((unsigned)m), mhi = __builtin_amdgcn_readfirstlane((unsigned)(m >> 32));
#pragma unroll
        for (int c = 0; c < N; c += GR) {
            const unsigned bits = (c < 32 ? (mlo >> (c & 31)) : (mhi >> (c & 31))) & ((1u << GR) - 1u);
            if (bits) {
                double2 t[GL];
#pragma unroll
                for (int u = 0; u < GL; ++u) t[u] = *reinterpret_cast<const double2*>(&db[c * 3 + 2 * u]);
                const double* tt = reinterpret_cast<const double*>(t);
#pragma unroll
                for (int i = 0; i < GR; ++i)
                    if (c + i < N) a[c + i] += tt[3 * i] * y0 + tt[3 * i + 1] * y1 + tt[3 * i + 2] * y2;
            }
        }
    }
    MVMC_WAVE_SYNC();
    // |g|^2 of the reduced gradient (= the Euler-space one: g is orthogonal to the null vectors), |g|_inf of the Euler-space gradient
    // B g_r, which is what SciPy tests
    if (!on) gj = 0.0;
    db[lane] = gj;
    MVMC_WAVE_SYNC();
    const double ge = expand(T, STAGE, lane, nae, db, refl);
    const double gg = uni(wave_sum_dpp(gj * gj)), ginf = uni(wave_max64(fabs(ge)));
    if (DBG) hh[MVMC_IK_DBG_G + lane] = ge;
    MVMC_WAVE_SYNC();
    if (lane == 0) { S.sc[0] = gg; S.sc[1] = ginf; }
    M1STAMP(4)
    if (ginf < gtol || !budget_left) { *mode_out = 0; return; }
    double scv, tauv;
    int ksteps;
    const int kk = uni(eightri::tridiag_krylov_w1<N>(a, gj, na, S.sv + SV_D, S.sv + SV_E, S.sv + SV_TAU, S.sv + SV_V0, S.tmp, S.tmp + 64,
                                                     &S.sc[4], scv, tauv, ksteps
#ifdef MVMC_TRI_PROFILE
                                                     , S.prof
#endif
                                                     ));
    // the reflectors, two rows to a register: the matrix registers die here, before the checks and the trust-region solve
    double pk[(N - 2) / 2];
    eightri::pack_reflectors<N>(a, pk);
    if (dump) { eightri::dump_reflectors<N>(pk, scv, ksteps, na, hh); if (lane < 55) hh[6400 + lane] = refl[lane]; }   // a solve that has rejected a trial before: see ik1_trf
    M1STAMP(5)
    bool ok = kk > 0;
    if (ok) ok = eightri::krylov_block_ok(S.sv + SV_D, S.sv + SV_E, kk, na, S.sc[6], S.sc[7], S.tmp, S.tmp + 64, S.tmp + 128,
                                          S.tmp + 192, S.sv + SV_WN);
    MVMC_WAVE_SYNC();
    M1STAMP(6)
    if (ok) {
        // ---- the trial step, in the tridiagonal basis ----
        const double beta0 = S.sc[4], tau0 = S.sc[5], pivmin = 1e-16 * S.sc[6] + 1e-300, coupling = S.sc[7];
        double* rh = S.tmp;
        double* cv = S.tmp + 64;
        rh[lane] = lane == 0 ? beta0 : 0.0;
        MVMC_WAVE_SYNC();
        double pred, step_norm;
        alpha = eightri::tr_solve_tri<false, N>(S.sv + SV_D, S.sv + SV_E, rh, kk, Delta, alpha, gg, pivmin, nullptr, nullptr, nullptr,
                                                nullptr, cv, &pred, &step_norm);
        MVMC_WAVE_SYNC();
        double c = lane < kk ? cv[lane] : 0.0;
        if (kk < na) {
            // component along the first null coordinate: keeps the step orthogonal to the null vector
            const double eta = coupling * wave_sum_dpp(lane < kk ? S.sv[SV_WN + lane] * c : 0.0);
            if (lane == kk) c = eta;
        }
        const double stepr = eightri::apply_q_packed<N>(pk, scv, tauv, S.sv + SV_V0, tau0, kk, na, c);
        if (lane == 0) { S.sc[2] = alpha; S.sc[3] = pred; S.sc[8] = step_norm; S.sc[11] = (double)kk; }
        MVMC_WAVE_SYNC();
        S.tmp[lane] = on ? stepr : 0.0;       // the step in reduced coordinates -> Euler space (the basis is orthonormal: same length)
        MVMC_WAVE_SYNC();
        const double stepj = expand(T, STAGE, lane, nae, S.tmp, refl);
        MVMC_WAVE_SYNC();
        ik1_trial_point(S, T, stage, nae, stepj);
        M1STAMP(3)
        *mode_out = 1;
        return;
    }
    // No clean split between range and null space (weakly observed directions, missing joints): the step is taken
    // in the eigenbasis of the tridiagonal matrix instead, with the numerically-null cluster removed -- the
    // eigensolver fallback in the T basis, where suf = V^T g = beta0 * (first components).
    //   unclean collapse:  T is complete (na rows);   clean collapse: the leading block plus its coupling row.
    // Its trials outlive this function: the reflectors go to the global scratch.
    if (!dump) { eightri::dump_reflectors<N>(pk, scv, ksteps, na, hh); if (lane < 55) hh[6400 + lane] = refl[lane]; }
    const int m = kk < 0 ? na : (kk < na ? kk + 1 : na);
    mvmc_gdouble* Zg = hh + 64 * NA1;
    eightri::tri_eigh_w1<N>(S.sv + SV_D, S.sv + SV_E, m, S.sv + SV_WN, Zg, S.tmp, S.tmp + 64, S.tmp + 128);
    const double suf = lane < m ? S.sc[4] * Zg[lane] : 0.0;
    MVMC_WAVE_SYNC();
    S.sv[SV_D + lane] = suf;   // d, e are dead: lam lives in the WN slot, suf in the D slot
    if (lane == 0) S.sc[10] = (double)m;
    MVMC_WAVE_SYNC();
    *mode_out = 2;
}

// A trial step of a model on the eigenbasis path (Delta, alpha -> S.xn and S.sc[2], [3], [8], [9])
__device__ __forceinline__ void ik1_finish_trial(Ik1Shared& S, const Ik1Tables& T, int stage, bool reduced, int na,
                                                 const mvmc_gdouble* __restrict__ hh, double stepj) {
    // stepj: lane j's component of the step in the model's coordinates.  Reduced model: back to Euler space with the reflectors the
    // model function parked in the global scratch (hh + 6400) next to its Householder vectors
    const int lane = threadIdx.x & 63;
    if (!reduced) { ik1_trial_point(S, T, stage, na, stepj); return; }
    const int nae = uni(T.na[stage]);
    MVMC_WAVE_SYNC();
    S.tmp[lane] = lane < na ? stepj : 0.0;
    if (lane < 55) S.tmp[192 + lane] = hh[6400 + lane];
    MVMC_WAVE_SYNC();
    const double se = arrow::expand(T, stage, lane, nae, S.tmp, S.tmp + 192);
    MVMC_WAVE_SYNC();
    ik1_trial_point(S, T, stage, nae, se);
}

__device__ __noinline__ void ik1_fallback_trial(Ik1Shared& S, const Ik1Tables& T, int stage, const mvmc_gdouble* __restrict__ hh,
                                                double Delta, double alpha, bool reduced) {
    MVMC_ASSUME_LDS(&S);
    MVMC_ASSUME_LDS(&T);
    const int lane = threadIdx.x & 63;
    const int na = uni(reduced ? (stage ? arrow::Dim<1>::NR : arrow::Dim<0>::NR) : T.na[stage]), mq = uni((int)S.sc[10]);
    const double gg = S.sc[0], tau0 = S.sc[5];
    double* cv = S.tmp + 64;
    double pred, step_norm;
    alpha = eightri::tr_solve_eig_w1(S.sv + SV_WN, S.sv + SV_D, mq, Delta, alpha, gg, cv, &pred, &step_norm);
    MVMC_WAVE_SYNC();
    const double c = eightri::eig_combine_w1(hh + 64 * NA1, mq, lane < mq ? cv[lane] : 0.0);
    const double stepj = eightri::apply_q_w1(hh, S.sv + SV_TAU, S.sv + SV_V0, tau0, mq, na, c);
    if (lane == 0) { S.sc[2] = alpha; S.sc[3] = pred; S.sc[8] = step_norm; }
    ik1_finish_trial(S, T, stage, reduced, na, hh, stepj);
}

// Another trial step (Delta, alpha) of a common-path model whose reflectors were dumped to the global scratch: the trust-region solve on
// the tridiagonal matrix that is still in LDS, Q from memory.
__device__ __noinline__ void ik1_retry_trial(Ik1Shared& S, const Ik1Tables& T, int stage, const mvmc_gdouble* __restrict__ hh,
                                             double Delta, double alpha, bool reduced) {
    MVMC_ASSUME_LDS(&S);
    MVMC_ASSUME_LDS(&T);
    const int lane = threadIdx.x & 63;
    const int na = uni(reduced ? (stage ? arrow::Dim<1>::NR : arrow::Dim<0>::NR) : T.na[stage]), kk = uni((int)S.sc[11]);
    const double gg = S.sc[0], beta0 = S.sc[4], tau0 = S.sc[5], pivmin = 1e-16 * S.sc[6] + 1e-300, coupling = S.sc[7];
    double* rh = S.tmp;
    double* cv = S.tmp + 64;
    rh[lane] = lane == 0 ? beta0 : 0.0;
    MVMC_WAVE_SYNC();
    double pred, step_norm;
    alpha = eightri::tr_solve_tri<false>(S.sv + SV_D, S.sv + SV_E, rh, kk, Delta, alpha, gg, pivmin, nullptr, nullptr, nullptr, nullptr,
                                         cv, &pred, &step_norm);
    MVMC_WAVE_SYNC();
    double c = lane < kk ? cv[lane] : 0.0;
    if (kk < na) {
        const double eta = coupling * wave_sum_dpp(lane < kk ? S.sv[SV_WN + lane] * c : 0.0);
        if (lane == kk) c = eta;
    }
    const double stepj = eightri::apply_q_w1(hh, S.sv + SV_TAU, S.sv + SV_V0, tau0, kk, na, c);
    if (lane == 0) { S.sc[2] = alpha; S.sc[3] = pred; S.sc[8] = step_norm; }
    ik1_finish_trial(S, T, stage, reduced, na, hh, stepj);
}

// ---------------------------------------------------------------------------------------------
// Two stage-1 models on one wave (mvmc_ik_pair.h; the SMALL layout of the chain kernel defines MVMC_IK_PAIR): what every build needs
// of it -- the pairing descriptor of a solve (dS = 0: not paired) and a wave's "away" bit.
// ---------------------------------------------------------------------------------------------
// bytes to the partner's solve block, doubles to its scratch, 0 / 1 = even / odd wave of the pair, the chain's void word (bit 3: a
// meeting timed out -- a bug, not a capacity: the results are void)
struct Ik1Pair { int dS, dhh, me; int32_t* ovf; };   // dS = 0: no pairing (every other kernel than the SMALL chain kernel)

__device__ __forceinline__ int* ik1_pair_mailbox(Ik1Shared& S, const Ik1Pair& P) {
    return &reinterpret_cast<Ik1Shared*>(reinterpret_cast<char*>(&S) + (P.me ? P.dS : 0))->pairw;
}
__device__ __forceinline__ int* ik1_pair_counter(Ik1Shared& S, const Ik1Pair& P) {
    return &reinterpret_cast<Ik1Shared*>(reinterpret_cast<char*>(&S) + (P.me ? 0 : P.dS))->pairw;
}
// the wave's "away" bit (one lane acts for the wave)
__device__ __forceinline__ void ik1_pair_away(Ik1Shared& S, const Ik1Pair& P, bool away) {
    if (P.dS == 0) return;
    if ((threadIdx.x & 63) == 0) {
        int* C = ik1_pair_counter(S, P);
        if (away) __hip_atomic_fetch_or(C, 1 << (16 + P.me), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_and(C, ~(1 << (16 + P.me)), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
#ifdef MVMC_IK_PAIR
#include "mvmc_ik_pair.h"
#endif

// ---------------------------------------------------------------------------------------------
// trf_no_bounds (trf.py:401-560) with x_scale = 1, linear loss, ftol = xtol = gtol = 1e-8 -- the loop of ik_trf
// on one wave.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void ik1_trf(Ik1Shared& S, const Ik1Tables& T, int stage, int max_nfev, mvmc_gdouble* __restrict__ hh,
                                        double* cost_out, int* nfev_out, int* njev_out, int* status_out, int* fallbacks_out, bool& dump,
                                        bool& fk_at_x, const Ik1Obs& O, const Ik1Pair& pair) {
    const int lane = threadIdx.x & 63;
    // wave-uniform state is pinned to scalar registers (uni)
    const int nfull = uni((stage == 0) ? 57 : 57 + T.n_side);
    const int na = uni(T.na[stage]);
    const double ftol = 1e-8, xtol = 1e-8, gtol = 1e-8;
    // The model in reduced coordinates (ik1_model_step_r) unless the skeleton is not the one its structures are written for or a
    // joint sits at gimbal lock (mode 3, decided before anything is touched): then the Euler-space model.  A solve stays with its
    // choice while it can: a rebuilt model (first rejected trial) asks again and gets the same answer from the same FK state.
    bool reduced = false;
    auto model_step = [&](bool budget_left, double Delta, double alpha, bool dump, bool may_pair) {
        // (the result comes back through the caller's stack on purpose: a call that is handed a pointer into its caller's frame is not
        // marked as a tail-call candidate, and only then does the compiler drop the callee-saved convention for this local function --
        // with it, the function's prologue saved and restored 58 vector registers that its caller does not even use)
        int mode = 3;
        if (uni(T.arrow_ok)) {
            if (stage == 0) {
                int code = -1;
#ifdef MVMC_IK_PAIR
                // a stage-1 model with evaluations left: together with the partner wave's, if it is at the same point (mvmc_ik_pair.h)
                if (pair.dS != 0 && may_pair && budget_left) { ik1_pair_sync(S, T, gtol, hh, Delta, alpha, dump, pair.dS, pair.dhh, pair.me, pair.ovf, &code); code = uni(code); }
#endif
#ifdef MVMC_PAIR_STATS   // diagnostic build (tools/pair_stats.py): meetings by outcome instead of the fallback count
                if (pair.dS != 0 && may_pair && budget_left) *fallbacks_out += code == 1 ? 1 : code == 2 ? 100 : 10000;
#endif
                if (code >= 1) mode = 1;
                else ik1_model_step_r<30, 0>(S, T, budget_left, gtol, hh, Delta, alpha, dump, &mode);
            }
            else ik1_model_step_r<40, 1>(S, T, budget_left, gtol, hh, Delta, alpha, dump, &mode);
        }
        reduced = uni(mode) != 3;
        if (!reduced) {
            if (na <= 40) ik1_model_step<40>(S, T, stage, budget_left, gtol, hh, Delta, alpha, dump, &mode);
            else ik1_model_step<50>(S, T, stage, budget_left, gtol, hh, Delta, alpha, dump, &mode);
        }
        return uni(mode);
    };
    double cost;
    { P1_T0 cost = uni(ik1_eval_nl(S, T, 0, stage, true, O.kps17, O.Pm, O.tg)); P1_ADD(0) }
    fk_at_x = true;    // (every other way out of the loop below leaves the FK state of x in LDS: the last evaluation was at the accepted point)
    int nfev = 1, njev = 0, status = -1;
    double Delta;
    {
        double xx = (lane < nfull) ? S.x[lane] * S.x[lane] : 0.0;
        if (lane + 64 < nfull) xx += S.x[lane + 64] * S.x[lane + 64];
        Delta = uni(sqrt(wave_sum(xx)));
    }
    if (Delta == 0.0) Delta = 1.0;
    double alpha = 0.0;
    if (stage == 0) ik1_pair_away(S, pair, false);   // this wave comes to the pair's meetings from here on
    while (true) {
        // the model at x (whose FK state and blocks are in LDS) and, unless it stops the iteration, the first trial from it
        int mode;
        if (stage == 0 && !(nfev < max_nfev)) ik1_pair_away(S, pair, true);   // (the last model of the stage is the gradient only: not paired)
        { P1_T0 mode = model_step(nfev < max_nfev, Delta, alpha, dump, true); P1_ADD(2) }
        bool dumped = dump;   // are this model's reflectors in the global scratch?
        ++njev;
        if (uni(S.sc[1]) < gtol) status = 1;
        if (status != -1 || nfev == max_nfev) break;
        if (mode == 2) ++*fallbacks_out;
        bool have_trial = mode == 1;
        double actual = -1.0, cost_new = cost;
        while (actual <= 0.0 && nfev < max_nfev) {
            if (!have_trial) {
                if (mode == 2) {
                    ik1_fallback_trial(S, T, stage, hh, Delta, alpha, reduced);
                } else if (dumped) {
                    ik1_retry_trial(S, T, stage, hh, Delta, alpha, reduced);
                } else {
                    // the first rejected trial of this solve: the reflectors died with the model's registers -- the state of x back
                    // into LDS (not an evaluation of the solver's budget: the same numbers again) and the model again, this time
                    // with its reflectors dumped to the global scratch, like every later model of the solve (solves that reject
                    // once tend to reject again: the cold ones, which set the length of a chain's first frame)
                    // (measured in round 4, bit-identical results: dumping only on demand -- every rejected first trial pays the rebuild,
                    // no model dumps pre-emptively -- takes 6.5 % off the launch's write traffic, 2.26 -> 2.11 GB, and costs 0.7 % of
                    // throughput; only the warm solves on demand: -1.6 % of the writes, no change in speed.  The dumps are not where
                    // the launch's 2.3 GB of writes come from; left as it was)
                    dump = dumped = true;
                    ik1_eval_nl(S, T, 0, stage, true, O.kps17, O.Pm, O.tg);
                    model_step(true, Delta, alpha, true, false);
                }
            }
            have_trial = false;
            alpha = uni(S.sc[2]);
            const double pred = uni(S.sc[3]), step_norm = uni(S.sc[8]), x_norm = uni(S.sc[9]);
            { P1_T0 cost_new = uni(ik1_eval_nl(S, T, 1, stage, true, O.kps17, O.Pm, O.tg)); P1_ADD(0) }
            ++nfev;
            if (!isfinite(cost_new)) { Delta = uni(0.25 * step_norm); continue; }
            actual = cost - cost_new;
            // update_tr_radius (common.py:222-245)
            double ratio;
            if (pred > 0.0) ratio = actual / pred;
            else if (pred == 0.0 && actual == 0.0) ratio = 1.0;
            else ratio = 0.0;
            double Delta_new = Delta;
            if (ratio < 0.25) Delta_new = 0.25 * step_norm;
            else if (ratio > 0.75 && step_norm > 0.95 * Delta) Delta_new = Delta * 2.0;
            // check_termination (common.py:705-717)
            const bool f_ok = (actual < ftol * cost) && (ratio > 0.25);
            const bool x_ok = step_norm < xtol * (xtol + x_norm);
            if (f_ok && x_ok) status = 4; else if (f_ok) status = 2; else if (x_ok) status = 3;
            if (status != -1) break;
            alpha = uni(alpha * (Delta / Delta_new));
            Delta = uni(Delta_new);
        }
        if (!(actual > 0.0)) { fk_at_x = false; break; }   // budget spent or stopped on a rejected trial: LDS holds the FK state of that trial
        if (lane < nfull) S.x[lane] = S.xn[lane];
        if (lane + 64 < nfull) S.x[lane + 64] = S.xn[lane + 64];
        MVMC_WAVE_SYNC();
        cost = cost_new;
        // the accepted point's FK state and blocks are still in LDS (the last evaluation was at xn)
        if (!(status == -1 && nfev < max_nfev)) break;
    }
    if (stage == 0) ik1_pair_away(S, pair, true);
    if (status == -1) status = 0;
    *cost_out = cost; *nfev_out = nfev; *njev_out = njev; *status_out = status;
}

// Cold start: DLT of the 18 keypoints + the reference's one-step post-optimisation; hips -> S.xn[0..6).  `views`: the problem's staged
// view block (ik1_stage_views: nv x 18 rows, then nv projection matrices) in the solve's global scratch -- chain heads only, one solve
// in sixteen on the benchmark protocol.
__device__ __noinline__ void ik1_cold_root(Ik1Shared& S, int nv, const double* views) {
    MVMC_ASSUME_LDS(&S);
    const int lane = threadIdx.x & 63;
    double X[3] = {0, 0, 0};
    const double* pose18 = views;
    const double* Pmv = views + nv * 54;
    if (lane < 18) dlt_obs_point(pose18, Pmv, nv, lane, 0.01, X);
    postopt::post_optimize_wave(X, pose18 + (lane < 18 ? lane : 0) * 3, 54, Pmv, nv, 18);
    if (lane == 11 || lane == 12)
        for (int c = 0; c < 3; ++c) S.xn[(lane - 11) * 3 + c] = X[c];
}

// Skeleton-derived tables (depth, ancestor masks, active columns and row masks of both stages).  ONE definition for the device (a wave
// builds them in LDS: lane = `lane`, a wave sync between the sections) and for the host (mvmc_chain_run builds them once per call and
// hands them to the chain kernel as a kernel argument -- built by every workgroup they cost 97 k cycles per frame, 2.4 % of a launch).
template <typename TB>
__host__ __device__ inline void ik1_tables_section(TB& T, const SkelDev& skarg, int section, int lane) {
    constexpr int obs_joint[NOBS] = MVMC_IK_SKEL_LIST;   // (= kIkSkel)
    const int n_side = skarg.n_side;
    if (section == 0) {
        if (lane < 18) {
            for (int k = 0; k < 3; ++k) T.dirs[lane * 3 + k] = skarg.dirs[lane][k];
            T.parents[lane] = (signed char)skarg.parents[lane];
            T.side_map[lane] = (signed char)skarg.side_map[lane];
            if (lane <= MVMC_N_SIDE) T.ref_side[lane] = skarg.ref_side[lane];
        }
    } else if (section == 1) {
        if (lane < 18) {
            int d = 0, m = 0;
            for (int a = T.parents[lane]; a >= 0; a = T.parents[a]) { ++d; m |= 1 << a; }
            T.depth[lane] = (signed char)d; T.anc[lane] = m;
            int sm = 0;
            for (int j = 0; j < 18; ++j) sm |= (T.side_map[j] == lane) ? (1 << j) : 0;
            T.smask[lane] = sm;
        }
    } else if (section == 2) {
        if (lane == 0) {
            int md = 0, moved = 0, lens = 0;  // joints whose rotation moves an observed joint; used length slots
            for (int j = 0; j < 18; ++j) md = T.depth[j] > md ? T.depth[j] : md;
            T.maxdepth = md;
            int nl = 0;
            for (int lev = 0; lev <= md; ++lev) {
                T.lev_start[lev] = (signed char)nl;
                for (int j = 0; j < 18; ++j)
                    if (T.depth[j] == lev) T.lev_list[nl++] = (signed char)j;
            }
            T.lev_start[md + 1] = (signed char)nl;
            T.n_side = n_side;
            for (int k = 0; k < NOBS; ++k) {
                const int K = obs_joint[k];
                moved |= T.anc[K];
                for (int j = K; j > 0; j = T.parents[j]) {
                    const double* d = &T.dirs[j * 3];
                    if (d[0] != 0.0 || d[1] != 0.0 || d[2] != 0.0) lens |= 1 << T.side_map[j];
                }
            }
            for (int st = 0; st < 2; ++st) {
                int n = 0;
                for (int c = 0; c < 3; ++c) { T.act[st][n] = c; T.colkind[st][n] = 0; T.cola[st][n] = 0; T.colc[st][n] = c; ++n; }
                for (int a = 0; a < 18; ++a)
                    if ((moved >> a) & 1)
                        for (int c = 0; c < 3 && n < NA1; ++c) {
                            T.act[st][n] = 3 + 3 * a + c; T.colkind[st][n] = 1; T.cola[st][n] = a; T.colc[st][n] = c; ++n;
                        }
                if (st == 1)
                    for (int s = 0; s < n_side && n < NA1; ++s)
                        if ((lens >> s) & 1) { T.act[st][n] = 57 + s; T.colkind[st][n] = 2; T.cola[st][n] = s; T.colc[st][n] = 0; ++n; }
                T.na[st] = n;
            }
        }
    } else if (section == 4) {
        if (lane == 0) {
            // the structures of the reduced coordinates: legs (hip, knee | ankle), arms (shoulder, elbow | wrist), spine + neck
            constexpr int want_par[18] = {-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 9, 10, 8, 12, 13, 8, 15, 15};
            constexpr int sja[5] = {1, 4, 9, 12, 7}, sjb[5] = {2, 5, 10, 13, 8}, stip[5] = {3, 6, 11, 14, -1};
            bool ok = n_side == 11 && T.na[0] == 39 && T.na[1] == 49;
            for (int j = 0; j < 18; ++j) ok = ok && T.parents[j] == want_par[j];
            for (int k = 0; k < NOBS; ++k) ok = ok && obs_joint[k] == (k < 7 ? k + 1 : k + 2);
            for (int s = 0; s < 5; ++s) { T.s_ja[s] = (signed char)sja[s]; T.s_jb[s] = (signed char)sjb[s]; T.s_tip[s] = (signed char)stip[s]; }
            int n = 0;
            for (int s = 0; s < 4; ++s)
                for (int c = 0; c < 4; ++c) { T.rja[n] = (unsigned char)s; T.rjc[n] = (unsigned char)c; ++n; }
            for (int c = 0; c < 3; ++c) { T.rja[n] = 15; T.rjc[n] = (unsigned char)c; ++n; }      // head: the nose's angles
            for (int c = 0; c < 3; ++c) { T.rja[n] = 0; T.rjc[n] = (unsigned char)c; ++n; }       // 19: translation
            for (int c = 0; c < 3; ++c) { T.rja[n] = 0; T.rjc[n] = (unsigned char)c; ++n; }       // 22: root rotation
            for (int c = 0; c < 5; ++c) { T.rja[n] = 4; T.rjc[n] = (unsigned char)c; ++n; }       // 25: spine + neck
            for (int l = 0; l < T.na[1]; ++l)
                if (T.colkind[1][l] == 2 && n < 40) { T.rja[n] = T.cola[1][l]; T.rjc[n] = 0; ++n; }   // 30: lengths
            ok = ok && n == 40;
            for (int l = 0; l < T.na[0]; ++l)      // stage 1's Euler columns are the first of stage 2's
                ok = ok && T.act[0][l] == T.act[1][l];
            {
                int n_len = 0;
                for (int l = 0; l < T.na[1]; ++l) {
                    const int kind = T.colkind[1][l], a = T.cola[1][l], c = T.colc[1][l];
                    int code = 0;
                    if (kind == 0) code = 19 + c;
                    else if (kind == 2) code = 30 + n_len++;
                    else if (a == 0) code = 22 + c;
                    else if (a == 15) code = 16 + c;
                    else {
                        int s_of = -1, comp = 0;
                        for (int s = 0; s < 5; ++s) {
                            if (sja[s] == a) { s_of = s; comp = c; }
                            if (sjb[s] == a) { s_of = s; comp = 3 + c; }
                        }
                        ok = ok && s_of >= 0;
                        code = -(1 + 8 * (s_of < 0 ? 0 : s_of) + comp);
                    }
                    T.e2r[l] = (signed char)code;
                }
                for (int k = 0; k < NOBS; ++k) {
                    const int K = obs_joint[k];
                    unsigned m = 0;
                    for (int i = 0; i < 10; ++i) {
                        const unsigned path = ((unsigned)T.anc[K] | (1u << K)) & (unsigned)T.smask[T.rja[30 + i]] & ~1u;
                        if (path) m |= 1u << i;
                    }
                    T.lenmask[k] = (unsigned short)m;
                }
            }
            T.arrow_ok = ok ? 1 : 0;
        }
    } else {
        if (lane < 2 * NOBS) {
            const int st = lane >> 4, k = lane & 15, K = obs_joint[k];
            unsigned long long m = 0;
            for (int col = 0; col < T.na[st]; ++col) {
                const int kind = T.colkind[st][col], a = T.cola[st][col];
                bool nz = kind == 0;
                if (kind == 1) nz = (T.anc[K] >> a) & 1;
                if (kind == 2)
                    for (int j = K; j > 0; j = T.parents[j]) nz |= T.side_map[j] == a;
                if (nz) m |= 1ull << col;
            }
            T.rowmask[st][k] = m;
        }
    }
}

template <typename TB>
__device__ __forceinline__ void ik1_build_tables(TB& T, const SkelDev& skarg) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int section = 0; section < 5; ++section) {
        ik1_tables_section(T, skarg, section, lane);
        MVMC_WAVE_SYNC();
    }
}

// the same tables on the host (every byte defined: the struct travels as a kernel argument)
inline void ik1_build_tables_host(Ik1Tables& T, const SkelDev& skarg) {
    unsigned char* bytes = reinterpret_cast<unsigned char*>(&T);
    for (size_t i = 0; i < sizeof(T); ++i) bytes[i] = 0;
    for (int section = 0; section < 5; ++section)
        for (int lane = 0; lane < 64; ++lane) ik1_tables_section(T, skarg, section, lane);
}

// The members of problem b into the member list (mq: pose index, mc: camera; LDS, room for vcap entries): returns the number of views
// used (at most vcap: more raise bit 0 of *ovf).  members (.., V): pose indices, -1 = none (holes allowed); n_valid >= 0: the row's first
// n_valid entries are the members (the rest of the row is undefined: the chain kernel's table).
__device__ __forceinline__ int ik1_rank_members(int* mq, unsigned short* mc, int vcap, const int32_t* __restrict__ members, int b, int V,
                                                int32_t* ovf, int n_valid, int C, int Pmax) {
    const int lane = threadIdx.x & 63;
    // lane v looks at member v (64 per pass); the valid ones are ranked by ballot and parked at their rank
    int nv = 0;
    if (n_valid >= 0) {
        nv = n_valid < 64 ? n_valid : 64;
        if (lane < nv && lane < vcap) { const int m = members[(size_t)b * V + lane]; mq[lane] = m; mc[lane] = (unsigned short)((m / Pmax) % C); }
    } else
    for (int v0 = 0; v0 < V; v0 += 64) {
        const int m = (v0 + lane < V) ? members[(size_t)b * V + v0 + lane] : -1;
        const unsigned long long have = __builtin_amdgcn_ballot_w64(m >= 0);
        const int rank = nv + __popcll(have & ((1ull << lane) - 1ull));
        if (m >= 0 && rank < vcap) { mq[rank] = m; mc[rank] = (unsigned short)((m / Pmax) % C); }
        nv += __popcll(have);
    }
    MVMC_WAVE_SYNC();
    nv = uni(nv);
    if (nv > vcap) { if (ovf && lane == 0) atomicOr(ovf, 1); nv = vcap; }
    return nv;
}

// The view block of a problem, staged for the cold start (global memory: the solve's scratch, free until the first model): 17 COCO rows
// + synthetic mid-spine (inverse_kinematics.py:339-348) per view, then the projection matrices.  Lane v writes view v, other lanes read
// it: an agent-scope release / acquire pair around the hand-over (the same wave, but the loads must not be served from stale L1 lines).
__device__ __forceinline__ void ik1_stage_views(mvmc_gdouble* views, int nv, const int* mq, const unsigned short* mc,
                                                const mvmc_gdouble* __restrict__ kps17, const mvmc_gdouble* __restrict__ Pmats) {
    const int lane = threadIdx.x & 63;
    if (lane < nv) {
        const mvmc_gdouble* kp = kps17 + (size_t)mq[lane] * 51;
        mvmc_gdouble* dst = views + lane * 54;
        for (int e = 0; e < 51; ++e) dst[e] = kp[e];
        for (int c = 0; c < 2; ++c) {
            const double mid_sh = 0.5 * (kp[5 * 3 + c] + kp[6 * 3 + c]);
            const double mid_hip = 0.5 * (kp[11 * 3 + c] + kp[12 * 3 + c]);
            dst[51 + c] = 0.5 * (mid_sh + mid_hip);
        }
        double sc = kp[5 * 3 + 2] * kp[6 * 3 + 2];
        sc *= kp[11 * 3 + 2] * kp[12 * 3 + 2];
        dst[53] = sc;
        const mvmc_gdouble* Pc = Pmats + (int)mc[lane] * 12;
        for (int e = 0; e < 12; ++e) views[nv * 54 + lane * 12 + e] = Pc[e];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// One solve on the calling wave (problem b); S is this wave's LDS block, (mq, mc) its member list (LDS, room for vcap views).  Used by
// ik1_kernel (one wave per workgroup) and by the chain kernel (four or eight waves per workgroup, one solve each).
// More members than vcap: the first vcap are used and bit 0 of *ovf is raised (the result is then not the reference's; with the chain
// kernel's pool and the stand-alone kernel's v_max list it cannot happen unless the caller's v_max is smaller than a cluster).
__device__ __forceinline__ void ik1_solve(Ik1Shared& S, int* mq, unsigned short* mc, int vcap, const Ik1Tables& T, const double* __restrict__ kps17,
                                          const double* __restrict__ Pmats, const int32_t* __restrict__ members, int b, int V,
                                          int C, int Pmax, const double* __restrict__ init, const uint8_t* __restrict__ cold,
                                          int nfev_cold, int nfev_warm, double* __restrict__ params_out,
                                          double* __restrict__ joints_out, double* __restrict__ info_out,
                                          double* __restrict__ scratch, int stage_mask, const double* __restrict__ targets3d,
                                          int32_t* ovf = nullptr, int n_valid = -1, int pair_dS = 0, int pair_dhh = 0, int pair_me = 0) {
    const int lane = threadIdx.x & 63;
    b = uni(b);
    mvmc_gdouble* hh = uni((mvmc_gdouble*)(scratch + (size_t)b * MVMC_IK_SCRATCH_DOUBLES));   // Householder vectors [0, 3200), eigenvectors [3200, 6400)
    double* info = uni(info_out ? info_out + (size_t)b * 8 : nullptr);
    const Ik1Obs O = {(const mvmc_gdouble*)kps17, (const mvmc_gdouble*)Pmats,
                      targets3d ? (const mvmc_gdouble*)(targets3d + (size_t)b * 72) : nullptr};
    if (lane == 0) {
        S.mq_off2 = (unsigned short)((reinterpret_cast<char*>(mq) - reinterpret_cast<char*>(&S)) >> 1);
        S.mc_off2 = (unsigned short)((reinterpret_cast<char*>(mc) - reinterpret_cast<char*>(&S)) >> 1);
    }
    // views of this problem (the reference only solves clusters with >= 2 views: motion_capture.py:927,940)
    if (targets3d == nullptr) {
        const int nv = ik1_rank_members(mq, mc, vcap, members, b, V, ovf, n_valid, C, Pmax);
        if (nv < 2) {
            const double nan = __longlong_as_double(0x7ff8000000000000LL);
            for (int i = lane; i < 68; i += 64) params_out[(size_t)b * 68 + i] = nan;
            if (lane < 54) joints_out[(size_t)b * 54 + lane] = nan;
            if (info && lane < 8) info[lane] = nan;
            return;
        }
        if (lane == 0) { S.nviews = nv; S.mode3d = 0; }
    } else {
        // 3-D-target mode: targets (B,18,4) in the observation row order (COCO-17 + mid-spine), read where they lie
        if (lane == 0) { S.nviews = 0; S.mode3d = 1; }
    }
    const int n_side = uni(T.n_side);
    MVMC_WAVE_SYNC();
    const int nv = uni(S.nviews);
    // ---- initial parameters ----
    // stage_mask bit 2: every problem starts from init (mvmc_ik_solve_stages); otherwise cold == NULL means all cold
    const bool is_cold = (stage_mask & 4) ? false : ((cold == nullptr) || uni((int)cold[b]) != 0);
    if (is_cold) {
        // root = midpoint of the triangulated (post-optimised) hips; zero angles; reference lengths
        // (inverse_kinematics.py:390-396 with triangulate(..., 0.01, post_optimize=True))
        ik1_stage_views(hh, nv, mq, mc, O.kps17, O.Pm);
        ik1_cold_root(S, nv, (const double*)hh);
        if (lane < 54) S.x[3 + lane] = 0.0;
        if (lane < n_side) { S.side[lane] = T.ref_side[lane]; S.x[57 + lane] = T.ref_side[lane]; }
        MVMC_WAVE_SYNC();
        if (lane < 3) S.x[lane] = 0.5 * (S.xn[lane] + S.xn[3 + lane]);
    } else {
        const double* p0 = init + (size_t)b * 68;
        for (int i = lane; i < 57 + n_side; i += 64) S.x[i] = p0[i];
        if (lane < n_side) S.side[lane] = p0[57 + lane];
    }
    MVMC_WAVE_SYNC();
#ifdef MVMC_IK_PROFILE
    if (lane < 8) S.prof[lane] = 0;
    const long long t_all = clock64();
    MVMC_WAVE_SYNC();
#endif
    const int max_nfev = uni(is_cold ? nfev_cold : nfev_warm);
    const Ik1Pair pair = {(targets3d || !uni(T.arrow_ok)) ? 0 : uni(pair_dS), uni(pair_dhh), uni(pair_me), uni(ovf)};
    double costs[2];
    int nfs[2], njs[2], sts[2], fallbacks = 0;
    bool dump = false, fk_final = false;   // fk_final: the FK state in LDS is that of the solution with stage-2 lengths
#pragma unroll 1
    for (int stage = 0; stage < 2; ++stage) {
        double c = 0.0; int nf = 0, nj = 0, st = 0;
        if ((stage_mask >> stage) & 1) {
            bool at_x = false;
            ik1_trf(S, T, stage, max_nfev, hh, &c, &nf, &nj, &st, &fallbacks, dump, at_x, O, pair);
            fk_final = at_x && stage == 1;
        }
        costs[stage] = c; nfs[stage] = nf; njs[stage] = nj; sts[stage] = st;
        MVMC_WAVE_SYNC();
    }
    // final FK at the solution -- unless the solver's last evaluation was at the accepted point (five of six warm solves): the same
    // function of the same numbers, already in LDS
    if (!fk_final) ik1_eval_nl(S, T, 0, 1, false, O.kps17, O.Pm, O.tg);
    for (int i = lane; i < 57 + n_side; i += 64) params_out[(size_t)b * 68 + i] = S.x[i];
    if (lane < 54) joints_out[(size_t)b * 54 + lane] = S.pos[lane];
    if (lane == 0) {
        if (info) {
            info[0] = costs[0]; info[1] = nfs[0]; info[2] = sts[0]; info[3] = costs[1]; info[4] = nfs[1]; info[5] = sts[1];
            info[6] = njs[0] + njs[1]; info[7] = fallbacks;
#ifdef MVMC_IK_PROFILE
            // diagnostic build only: cycle counts instead of the costs (the slots tools/ik_chain_profile.py reads)
            info[0] = (double)S.prof[0]; info[3] = 0.0; info[2] = (double)S.prof[2];
            info[5] = (double)(clock64() - t_all); info[7] = (double)S.prof[3];
            info[1] = (double)S.prof[4]; info[4] = (double)S.prof[5]; info[6] = (double)S.prof[6];
#endif
        }
    }
}

__global__ void __launch_bounds__(64, MVMC_SMALL_WPS)
ik1_kernel(SkelDev skarg, const double* __restrict__ kps17, const double* __restrict__ Pmats,
           const int32_t* __restrict__ members, int B, int V, int vcap, int C, int Pmax, const double* __restrict__ init,
           const uint8_t* __restrict__ cold, int nfev_cold, int nfev_warm, double* __restrict__ params_out,
           double* __restrict__ joints_out, double* __restrict__ info_out, double* __restrict__ scratch, int stage_mask,
           const double* __restrict__ targets3d, int32_t* __restrict__ overflow) {
    __shared__ Ik1Shared S;
    __shared__ Ik1Tables T;
    extern __shared__ __attribute__((aligned(16))) int ik1_members[];   // the member list: vcap pose indices, then vcap cameras (uint16)
    ik1_build_tables(T, skarg);
    ik1_solve(S, ik1_members, reinterpret_cast<unsigned short*>(ik1_members + vcap), vcap, T, kps17, Pmats, members, blockIdx.x, V, C, Pmax, init, cold, nfev_cold, nfev_warm, params_out,
              joints_out, info_out, scratch, stage_mask, targets3d, overflow ? overflow + blockIdx.x : nullptr);
}

#ifndef MVMC_DEVICE_ONLY
// ---------------------------------------------------------------------------------------------
// Diagnostic (mvmc_debug_ik_model_step): ONE trust-region model + ONE trial step of the production solver from a caller-given
// (x, Delta, alpha) -- the unit in which the reference's recorded iterates are compared (tests/test_gpu_ik_trf_traces.py).  The
// functions are the solver's own (evaluation, model in reduced coordinates or its Euler-space form, fallback trial); the DBG
// instances of the model functions differ in one store (the gradient).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64, MVMC_SMALL_WPS)
ik1_step_kernel(SkelDev skarg, const double* __restrict__ kps17, const double* __restrict__ Pmats, const int32_t* __restrict__ members,
                int V, int vcap, int C, int Pmax, const double* __restrict__ params, int stage, const double* __restrict__ Delta_in,
                const double* __restrict__ alpha_in, double* __restrict__ out, double* __restrict__ scratch) {
    __shared__ Ik1Shared S;
    __shared__ Ik1Tables T;
    extern __shared__ __attribute__((aligned(16))) int ik1_members[];
    const int lane = threadIdx.x & 63, b = blockIdx.x;
    ik1_build_tables(T, skarg);
    mvmc_gdouble* hh = uni((mvmc_gdouble*)(scratch + (size_t)b * MVMC_IK_SCRATCH_DOUBLES));
    double* o = out + (size_t)b * MVMC_IK_STEP_OUT_DOUBLES;
    for (int i = lane; i < MVMC_IK_STEP_OUT_DOUBLES; i += 64) o[i] = 0.0;
    int* mq = ik1_members;
    unsigned short* mc = reinterpret_cast<unsigned short*>(ik1_members + vcap);
    const Ik1Obs O = {(const mvmc_gdouble*)kps17, (const mvmc_gdouble*)Pmats, nullptr};
    if (lane == 0) {
        S.mq_off2 = (unsigned short)((reinterpret_cast<char*>(mq) - reinterpret_cast<char*>(&S)) >> 1);
        S.mc_off2 = (unsigned short)((reinterpret_cast<char*>(mc) - reinterpret_cast<char*>(&S)) >> 1);
    }
    const int nv = ik1_rank_members(mq, mc, vcap, members, b, V, nullptr, -1, C, Pmax);
    if (nv < 2) { if (lane == 0) o[6] = -1.0; return; }
    if (lane == 0) { S.nviews = nv; S.mode3d = 0; }
    const int n_side = uni(T.n_side);
    MVMC_WAVE_SYNC();
    const double* p0 = params + (size_t)b * 68;
    for (int i = lane; i < 57 + n_side; i += 64) S.x[i] = p0[i];
    if (lane < n_side) S.side[lane] = p0[57 + lane];
    MVMC_WAVE_SYNC();
    const int nfull = (stage == 0) ? 57 : 57 + n_side;
    const int na = uni(T.na[stage]);
    const double Delta = Delta_in[b], alpha0 = alpha_in[b], gtol = 1e-8;
    const double cost = uni(ik1_eval_nl(S, T, 0, stage, true, O.kps17, O.Pm, O.tg));
    int mode = 3;
    bool reduced = false;
    if (uni(T.arrow_ok)) {
        if (stage == 0) ik1_model_step_r<30, 0, true>(S, T, true, gtol, hh, Delta, alpha0, false, &mode);
        else ik1_model_step_r<40, 1, true>(S, T, true, gtol, hh, Delta, alpha0, false, &mode);
    }
    reduced = uni(mode) != 3;
    if (!reduced) {
        if (na <= 40) ik1_model_step<40, true>(S, T, stage, true, gtol, hh, Delta, alpha0, false, &mode);
        else ik1_model_step<50, true>(S, T, stage, true, gtol, hh, Delta, alpha0, false, &mode);
    }
    mode = uni(mode);
    if (mode == 2) ik1_fallback_trial(S, T, stage, hh, Delta, alpha0, reduced);
    MVMC_WAVE_SYNC();
    if (lane < na) o[8 + T.act[stage][lane]] = hh[MVMC_IK_DBG_G + lane];
    double cost_new = 0.0;
    if (mode != 0) {
        for (int i = lane; i < nfull; i += 64) o[80 + i] = S.xn[i] - S.x[i];
        for (int i = lane; i < nfull; i += 64) o[160 + i] = S.xn[i];
        MVMC_WAVE_SYNC();
        cost_new = uni(ik1_eval_nl(S, T, 1, stage, false, O.kps17, O.Pm, O.tg));
    }
    if (lane == 0) {
        o[0] = cost; o[1] = S.sc[1];
        if (mode != 0) { o[2] = S.sc[2]; o[3] = S.sc[3]; o[4] = S.sc[8]; o[5] = cost_new; o[7] = mode == 1 ? S.sc[11] : S.sc[10]; }
        o[6] = (double)(mode + (reduced ? 0 : 4));
    }
}
#endif

}  // namespace

#ifndef MVMC_DEVICE_ONLY   // (mvmc_chain.hip includes the device code above)
static int mvmc_ik1_launch(const SkelDev& sk, const double* kps17, const double* Pmats, const int32_t* members, int n_problems,
                           int v_max, int n_views, int p_max, const double* init_params, const uint8_t* cold, int max_nfev_cold,
                           int max_nfev_warm, double* params_out, double* joints_out, double* info_out, double* scratch,
                           int stage_mask, const double* targets3d, hipStream_t stream) {
    // the view block holds every member a problem can have (v_max columns), so no cluster is ever cut short here
    const int vcap = v_max < 1 ? 1 : v_max;
    if (vcap > 64 || n_views > 65535) return MVMC_ERR_UNSUPPORTED;   // (a lane per view in the member list; the list holds cameras as uint16)
    const size_t lds = ((size_t)vcap * 6 + 15) / 16 * 16;   // the member list: an int and a short per view
    hipLaunchKernelGGL(ik1_kernel, dim3(n_problems), dim3(64), lds, stream, sk, kps17, Pmats, members, n_problems, v_max, vcap,
                       n_views, p_max, init_params, cold, max_nfev_cold, max_nfev_warm, params_out, joints_out, info_out,
                       scratch, stage_mask, targets3d, (int32_t*)nullptr);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}

extern "C" int mvmc_ik_solve(const mvmcSkeleton* skel_host, const double* kps17, const double* Pmats,
                             const int32_t* members, int n_problems, int v_max, int n_views, int p_max,
                             const double* init_params, const uint8_t* cold, int max_nfev_cold, int max_nfev_warm,
                             double* params_out, double* joints_out, double* info_out, double* scratch,
                             mvmcStream_t stream) {
    if (!skel_host || !kps17 || !Pmats || !members || !params_out || !joints_out || !scratch) return MVMC_ERR_ARG;
    if (v_max <= 0 || n_views <= 0 || p_max <= 0 || max_nfev_cold < 1 || max_nfev_warm < 1) return MVMC_ERR_ARG;
    if (cold && !init_params) return MVMC_ERR_ARG;
    if (n_problems <= 0) return n_problems == 0 ? MVMC_OK : MVMC_ERR_ARG;
    SkelDev sk;
    if (!skel_to_dev(skel_host, &sk)) return MVMC_ERR_ARG;
    if (sk.n_side != MVMC_N_SIDE) return MVMC_ERR_UNSUPPORTED;  // the solver is sized for 57 + 11 parameters
    return mvmc_ik1_launch(sk, kps17, Pmats, members, n_problems, v_max, n_views, p_max, init_params, init_params ? cold : nullptr,
                           max_nfev_cold, max_nfev_warm, params_out, joints_out, info_out, scratch, 3, nullptr, (hipStream_t)stream);
}

// Single stages of PoseSolver.solve and the 3-D-target variants; see include/mvmc.h
extern "C" int mvmc_ik_solve_stages(const mvmcSkeleton* skel_host, const double* kps17, const double* Pmats,
                                    const int32_t* members, const double* targets3d, int n_problems, int v_max, int n_views,
                                    int p_max, const double* init_params, int stage_mask, int max_nfev,
                                    double* params_out, double* joints_out, double* info_out, double* scratch,
                                    mvmcStream_t stream) {
    if (!skel_host || !init_params || !params_out || !joints_out || !scratch) return MVMC_ERR_ARG;
    if (stage_mask < 1 || stage_mask > 3 || max_nfev < 1) return MVMC_ERR_ARG;
    if (!targets3d && (!kps17 || !Pmats || !members || v_max <= 0 || n_views <= 0 || p_max <= 0)) return MVMC_ERR_ARG;
    if (n_problems <= 0) return n_problems == 0 ? MVMC_OK : MVMC_ERR_ARG;
    SkelDev sk;
    if (!skel_to_dev(skel_host, &sk)) return MVMC_ERR_ARG;
    if (sk.n_side != MVMC_N_SIDE) return MVMC_ERR_UNSUPPORTED;
    // every problem is "warm": it starts from init_params with the one evaluation budget
    return mvmc_ik1_launch(sk, kps17, Pmats, members, n_problems, targets3d ? 1 : v_max, targets3d ? 1 : n_views,
                           targets3d ? 1 : p_max, init_params, /*cold=*/nullptr, max_nfev, max_nfev, params_out, joints_out,
                           info_out, scratch, stage_mask | 4, targets3d, (hipStream_t)stream);
}

extern "C" int mvmc_debug_ik_model_step(const mvmcSkeleton* skel_host, const double* kps17, const double* Pmats, const int32_t* members,
                                        int n_problems, int v_max, int n_views, int p_max, const double* params, int stage,
                                        const double* Delta, const double* alpha0, double* out, double* scratch, mvmcStream_t stream) {
    if (!skel_host || !kps17 || !Pmats || !members || !params || !Delta || !alpha0 || !out || !scratch) return MVMC_ERR_ARG;
    if (v_max <= 0 || n_views <= 0 || p_max <= 0 || stage < 0 || stage > 1) return MVMC_ERR_ARG;
    if (n_problems <= 0) return n_problems == 0 ? MVMC_OK : MVMC_ERR_ARG;
    SkelDev sk;
    if (!skel_to_dev(skel_host, &sk)) return MVMC_ERR_ARG;
    if (sk.n_side != MVMC_N_SIDE) return MVMC_ERR_UNSUPPORTED;
    if (v_max > 64 || n_views > 65535) return MVMC_ERR_UNSUPPORTED;
    const size_t lds = ((size_t)v_max * 6 + 15) / 16 * 16;
    hipLaunchKernelGGL(ik1_step_kernel, dim3(n_problems), dim3(64), lds, (hipStream_t)stream, sk, kps17, Pmats, members, v_max, v_max,
                       n_views, p_max, params, stage, Delta, alpha0, out, scratch);
    MVMC_CHECK_LAUNCH();
    return MVMC_OK;
}
#endif  // MVMC_DEVICE_ONLY
