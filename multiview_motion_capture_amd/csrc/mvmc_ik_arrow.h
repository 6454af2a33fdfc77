// Reduced coordinates of the IK trust-region model (one wave per solve; used by ik1_model_step_r in mvmc_ik1.hip).
//
// Nine null directions of the Euler-angle Jacobian are STRUCTURAL, known in closed form from the FK state: a limb (hip -> knee ->
// ankle, shoulder -> elbow -> wrist) is seen through two points, so of the six Euler rates of its two joints only four move anything
// -- the twist of the lower joint about its own bone moves nothing, and the twist of the upper joint about ITS bone is undone by the
// opposite rotation of the lower joint about the same line (2 x 4 limbs); likewise the twist of the spine about the spine -> neck
// bone against the neck's (1).  SciPy's step (common.py:57-168) is the minimum-norm solution, i.e. orthogonal to all of them, so
// the sub-problem can be posed in an orthonormal basis of their complement without changing it: per limb a 6 x 4 basis, for spine +
// neck 6 x 5, built from two Householder reflectors that map the null vectors onto the last coordinates (no pivoting, no branches).
// 30 / 40 reduced columns instead of 39 / 49; column order: the limbs LL 0-3, RL 4-7, LA 8-11, RA 12-15, the head 16-18 (the
// nose's three angles), then translation 19-21, root rotation 22-24, spine + neck 25-29, the side lengths 30-39 (stage 2).
//
// In these coordinates the matrix is an ARROW (the limbs couple with the trunk and not with one another) and, when every joint is
// observed, positive definite.  A block Cholesky factorisation per alpha along that sparsity was built and measured (round 3): ~650
// fused multiply-adds per lane and factorisation against ~7,000 for the tridiagonalisation -- but SciPy's Newton iteration on alpha
// needs 3.7 factorisations per model, each a chain of 4 + 21 dependent rounds in which at most 21 of the 64 lanes work, and a wave
// instruction costs the same with 21 lanes as with 64: 13 k instructions per model against 7 k, a warm solve 1.82 M cycles against
// 1.59 M.  Dropped; the reduced basis is kept with the dense tridiagonalisation, which touches all rows once per model.
#pragma once

namespace arrow {

constexpr int NLIMB = 19;                      // reduced limb columns
template <int STAGE> struct Dim { static constexpr int NT = STAGE ? 21 : 11, NR = NLIMB + NT, NL = 4 + NT; };

// E_j^-1 (Gp^T v): the Euler rates of joint j that produce the angular velocity v (world frame); *c1 = cos(e_y) (0 = gimbal lock)
__device__ __forceinline__ void euler_rates(const double* hs, const double* Gp, const double v[3], double n[3], double* c1_out) {
    const double s0 = 2.0 * hs[0] * hs[1], c0 = hs[1] * hs[1] - hs[0] * hs[0];
    const double s1 = 2.0 * hs[2] * hs[3], c1 = hs[3] * hs[3] - hs[2] * hs[2];
    const double l0 = Gp[0] * v[0] + Gp[3] * v[1] + Gp[6] * v[2];
    const double l1 = Gp[1] * v[0] + Gp[4] * v[1] + Gp[7] * v[2];
    const double l2 = Gp[2] * v[0] + Gp[5] * v[1] + Gp[8] * v[2];
    n[2] = (c0 * l2 - s0 * l1) / c1;
    n[1] = c0 * l1 + s0 * l2;
    n[0] = l0 - s1 * n[2];
    *c1_out = c1;
}
// the world-frame angular velocity of Euler rates b[0..3) of joint j: Gp (b0 + b2 s1, b1 c0 - b2 s0 c1, b1 s0 + b2 c0 c1)
__device__ __forceinline__ void omega_of(const double* hs, const double* Gp, bool root, const double b[3], double w[3]) {
    const double s0 = 2.0 * hs[0] * hs[1], c0 = hs[1] * hs[1] - hs[0] * hs[0];
    const double s1 = 2.0 * hs[2] * hs[3], c1 = hs[3] * hs[3] - hs[2] * hs[2];
    const double l0 = b[0] + b[2] * s1, l1 = b[1] * c0 - b[2] * s0 * c1, l2 = b[1] * s0 + b[2] * c0 * c1;
    if (root) { w[0] = l0; w[1] = l1; w[2] = l2; return; }
    w[0] = Gp[0] * l0 + Gp[1] * l1 + Gp[2] * l2;
    w[1] = Gp[3] * l0 + Gp[4] * l1 + Gp[5] * l2;
    w[2] = Gp[6] * l0 + Gp[7] * l1 + Gp[8] * l2;
}
// The reflectors of structure s (0 .. 3: the limbs, two null vectors; 4: spine + neck, one) from the FK state: out[0..6) = u1,
// out[6..11) = u2 (zero: no second reflector).  Returns true at gimbal lock (or a non-finite result): not applicable.
template <typename Tables>
__device__ __forceinline__ bool structure_reflectors(const double* pos, const double* hs, const double* Rg, const Tables& T, int s,
                                                     double* out) {
    const int ja = T.s_ja[s], jb = T.s_jb[s], tip = T.s_tip[s];
    double bd[3], n2[6], c1a, c1b;
    {
        const double b0 = pos[jb * 3] - pos[ja * 3], b1 = pos[jb * 3 + 1] - pos[ja * 3 + 1], b2 = pos[jb * 3 + 2] - pos[ja * 3 + 2];
        const double inv = 1.0 / sqrt(b0 * b0 + b1 * b1 + b2 * b2);
        bd[0] = b0 * inv; bd[1] = b1 * inv; bd[2] = b2 * inv;
    }
    // n2: the upper joint's twist about the bone to the lower joint, against the lower joint's rotation about the same line
    euler_rates(&hs[ja * 4], &Rg[T.parents[ja] * 9], bd, n2, &c1a);
    euler_rates(&hs[jb * 4], &Rg[T.parents[jb] * 9], bd, n2 + 3, &c1b);
    n2[3] = -n2[3]; n2[4] = -n2[4]; n2[5] = -n2[5];
    bool bad = !(fabs(c1a) > 1e-6) || !(fabs(c1b) > 1e-6);
    double u1[6], u2[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (tip >= 0) {
        // n1: the lower joint's twist about its own bone
        double cd[3], n1[3];
        const double c0 = pos[tip * 3] - pos[jb * 3], c1 = pos[tip * 3 + 1] - pos[jb * 3 + 1], c2 = pos[tip * 3 + 2] - pos[jb * 3 + 2];
        const double inv = 1.0 / sqrt(c0 * c0 + c1 * c1 + c2 * c2);
        cd[0] = c0 * inv; cd[1] = c1 * inv; cd[2] = c2 * inv;
        euler_rates(&hs[jb * 4], &Rg[T.parents[jb] * 9], cd, n1, &c1b);
        // H1: (0, 0, 0, n1) -> e6
        const double sg = sqrt(n1[0] * n1[0] + n1[1] * n1[1] + n1[2] * n1[2]);
        u1[0] = 0.0; u1[1] = 0.0; u1[2] = 0.0; u1[3] = n1[0]; u1[4] = n1[1]; u1[5] = n1[2] + (n1[2] >= 0.0 ? sg : -sg);
        const double i1 = 1.0 / sqrt(u1[3] * u1[3] + u1[4] * u1[4] + u1[5] * u1[5]);
        u1[3] *= i1; u1[4] *= i1; u1[5] *= i1;
        // H2: the first five components of H1 n2 -> e5
        double d1 = 0.0;
        for (int i = 0; i < 6; ++i) d1 += u1[i] * n2[i];
        double v[5], q = 0.0;
        for (int i = 0; i < 5; ++i) { v[i] = n2[i] - 2.0 * d1 * u1[i]; q += v[i] * v[i]; }
        const double s2 = sqrt(q);
        v[4] += v[4] >= 0.0 ? s2 : -s2;
        double qq = 0.0;
        for (int i = 0; i < 5; ++i) qq += v[i] * v[i];
        const double i2 = 1.0 / sqrt(qq);
        for (int i = 0; i < 5; ++i) u2[i] = v[i] * i2;
    } else {
        // one null vector: H1: n2 -> e6
        double q = 0.0;
        for (int i = 0; i < 6; ++i) q += n2[i] * n2[i];
        const double sg = sqrt(q);
        for (int i = 0; i < 6; ++i) u1[i] = n2[i];
        u1[5] += n2[5] >= 0.0 ? sg : -sg;
        double qq = 0.0;
        for (int i = 0; i < 6; ++i) qq += u1[i] * u1[i];
        const double i1 = 1.0 / sqrt(qq);
        for (int i = 0; i < 6; ++i) u1[i] *= i1;
    }
    for (int i = 0; i < 6; ++i) out[i] = u1[i];
    for (int i = 0; i < 5; ++i) out[6 + i] = u2[i];
    return bad || !(u1[5] == u1[5]) || !(u2[4] == u2[4]);
}
// z (6) <- H1 (H2 z[0..5) (+) z[5]) with the reflectors u1 (6), u2 (5) of a structure (u2 = 0: no second reflector)
__device__ __forceinline__ void apply_reflectors(const double* u, double z[6]) {
    double s2 = 0.0;
#pragma unroll
    for (int i = 0; i < 5; ++i) s2 += u[6 + i] * z[i];
#pragma unroll
    for (int i = 0; i < 5; ++i) z[i] -= 2.0 * s2 * u[6 + i];
    double s1 = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) s1 += u[i] * z[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) z[i] -= 2.0 * s1 * u[i];
}
// component of the Euler-space vector B v_r that belongs to Euler lane `lane` (v_r: reduced vector in LDS)
template <typename Tables>
__device__ __forceinline__ double expand(const Tables& T, int stage, int lane, int na, const double* vr, const double* refl) {
    if (lane >= na) return 0.0;
    (void)stage;
    const int code = T.e2r[lane];
    if (code >= 0) return vr[code];
    const int s = (-code - 1) >> 3, comp = (-code - 1) & 7;
    const int base = s < 4 ? 4 * s : NLIMB + 6, nc = s < 4 ? 4 : 5;
    double z[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) z[i] = i < nc ? vr[base + i] : 0.0;
    apply_reflectors(refl + s * 11, z);
    double r = z[0];
#pragma unroll
    for (int i = 1; i < 6; ++i) r = comp == i ? z[i] : r;
    return r;
}

}  // namespace arrow
