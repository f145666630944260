// cfnmpc_model.hpp -- Crazyflie 13-state / 4-input model on the device (gfx950, FP64).
//
// Restates, for the GPU, the continuous dynamics of
//   crazyflie_controller/scripts/crazyflie_full_model/export_ode_model.py:33-102
// and the structure of its RK4 sensitivities.  State order (acados_mpc.cpp:117-131)
//   x = [ p(0..2) | q(3..6: w x y z) | v(7..9, body frame) | w(10..12, body rates) ],  u = kRPM(4).
//
// Nothing depends on p, q' depends on (q,w), v' on (q,v,w,u), w' on (w,u), p' on (q,v):
// influence graph  w -> q -> v -> p.  Hence d Phi / d x of ANY explicit RK scheme has the
// block pattern (rows x cols, blocks p,q,v,w)
//        p  q  v  w
//    p [ I  *  *  * ]
//    q [ 0  *  0  * ]        97 stored entries + 3 unit entries instead of 169,
//    v [ 0  *  *  * ]
//    w [ 0  0  0  * ]
// which the linearisation kernel writes and the Riccati kernels read in compact form.
#pragma once
#include <hip/hip_runtime.h>

namespace cfn {

constexpr int NX = 13, NU = 4, NY = 17;

// export_ode_model.py:34-42
constexpr double G0 = 9.8066, MQ = 33e-3, IXX = 1.395e-5, IYY = 1.395e-5, IZZ = 2.173e-5,
                 CD = 7.9379e-06, CT = 3.25e-4, ARM = 65e-3 / 2;
constexpr double KWX = -(IZZ - IYY) / IXX, KWY = -(IXX - IZZ) / IYY, KWZ = -(IYY - IXX) / IZZ;
constexpr double KT = CT / MQ, KA = -CT * ARM / IXX, KB = -CT * ARM / IYY, KC = -CD / IZZ;

// ---- compact pattern of A = d Phi / d x -------------------------------------------------
__host__ __device__ constexpr int blk(int i) { return i < 3 ? 0 : (i < 7 ? 1 : (i < 10 ? 2 : 3)); }
// 0: structural zero, 1: unit entry (p-block diagonal), 2: stored
__host__ __device__ constexpr int a_kind(int r, int c) {
    const int br = blk(r), bc = blk(c);
    if (bc == 0) return (r == c) ? 1 : 0;
    if (bc == 1) return (br <= 2) ? 2 : 0;
    if (bc == 2) return (br == 0 || br == 2) ? 2 : 0;
    return 2;
}
__host__ __device__ constexpr int a_idx(int r, int c) {  // row-major rank among stored entries
    int n = 0;
    for (int i = 0; i < NX; i++)
        for (int j = 0; j < NX; j++) {
            if (i == r && j == c) return n;
            if (a_kind(i, j) == 2) n++;
        }
    return n;
}
constexpr int A_NNZ = a_idx(NX - 1, NX - 1) + 1;  // 97
static_assert(A_NNZ == 97, "pattern of dPhi/dx");

// packed symmetric 13x13 (upper, row-major): index of (i,j), any order of i,j
__host__ __device__ constexpr int sidx(int i, int j) {
    const int a = i < j ? i : j, b = i < j ? j : i;
    return a * NX - (a * (a - 1)) / 2 + (b - a);
}
constexpr int S_NNZ = 91;
// packed symmetric 4x4
__host__ __device__ constexpr int s4(int i, int j) {
    const int a = i < j ? i : j, b = i < j ? j : i;
    return a * 4 - (a * (a - 1)) / 2 + (b - a);
}

// ---- continuous dynamics ----------------------------------------------------------------
__device__ __forceinline__ void f_expl(const double* __restrict__ x, const double* __restrict__ u,
                                       double* __restrict__ dx) {
    const double q1 = x[3], q2 = x[4], q3 = x[5], q4 = x[6];
    const double vbx = x[7], vby = x[8], vbz = x[9];
    const double wx = x[10], wy = x[11], wz = x[12];
    const double w1 = u[0], w2 = u[1], w3 = u[2], w4 = u[3];
    dx[0] = vbx * (2 * q1 * q1 + 2 * q2 * q2 - 1) - vby * (2 * q1 * q4 - 2 * q2 * q3) + vbz * (2 * q1 * q3 + 2 * q2 * q4);
    dx[1] = vby * (2 * q1 * q1 + 2 * q3 * q3 - 1) + vbx * (2 * q1 * q4 + 2 * q2 * q3) - vbz * (2 * q1 * q2 - 2 * q3 * q4);
    dx[2] = vbz * (2 * q1 * q1 + 2 * q4 * q4 - 1) - vbx * (2 * q1 * q3 - 2 * q2 * q4) + vby * (2 * q1 * q2 + 2 * q3 * q4);
    dx[3] = -(q2 * wx) / 2 - (q3 * wy) / 2 - (q4 * wz) / 2;
    dx[4] = (q1 * wx) / 2 - (q4 * wy) / 2 + (q3 * wz) / 2;
    dx[5] = (q4 * wx) / 2 + (q1 * wy) / 2 - (q2 * wz) / 2;
    dx[6] = (q2 * wy) / 2 - (q3 * wx) / 2 + (q1 * wz) / 2;
    dx[7] = vby * wz - vbz * wy + G0 * (2 * q1 * q3 - 2 * q2 * q4);
    dx[8] = vbz * wx - vbx * wz - G0 * (2 * q1 * q2 + 2 * q3 * q4);
    // (rotor terms and gyroscopic couplings with the constants folded at compile time -- KT = Ct / mq, KA = -Ct l / Ixx, ...,
    //  KWX = -(Izz - Iyy) / Ixx, ...: the same products jvp() differentiates; a division by mq / Ixx / Iyy / Izz per
    //  evaluation cost k_linearise 16 FP64 divisions per shooting interval)
    const double s1 = w1 * w1, s2 = w2 * w2, s3 = w3 * w3, s4 = w4 * w4;
    dx[9] = vbx * wy - vby * wx - G0 * (2 * q1 * q1 + 2 * q4 * q4 - 1) + KT * (s1 + s2 + s3 + s4);
    dx[10] = KA * (s1 + s2 - s3 - s4) + KWX * (wy * wz);
    dx[11] = KB * (s1 - s2 - s3 + s4) + KWY * (wx * wz);
    dx[12] = KC * (s1 - s2 + s3 - s4) + KWZ * (wx * wy);
}

// Point data of df/dx at one RK stage point that is shared by all sensitivity columns:
// rotation matrix R(q) (= dp'/dv) and dp'/dq; everything else is linear in (q,v,w).
struct JacPoint {
    double q[4], v[3], w[3];
    double R[9];
    double Jpq[12];
};

__device__ __forceinline__ void jac_point(const double* __restrict__ x, JacPoint& J) {
    const double q1 = x[3], q2 = x[4], q3 = x[5], q4 = x[6];
    const double vx = x[7], vy = x[8], vz = x[9];
    J.q[0] = q1; J.q[1] = q2; J.q[2] = q3; J.q[3] = q4;
    J.v[0] = vx; J.v[1] = vy; J.v[2] = vz;
    J.w[0] = x[10]; J.w[1] = x[11]; J.w[2] = x[12];
    J.R[0] = 2 * q1 * q1 + 2 * q2 * q2 - 1; J.R[1] = -(2 * q1 * q4 - 2 * q2 * q3); J.R[2] = 2 * q1 * q3 + 2 * q2 * q4;
    J.R[3] = 2 * q1 * q4 + 2 * q2 * q3; J.R[4] = 2 * q1 * q1 + 2 * q3 * q3 - 1; J.R[5] = -(2 * q1 * q2 - 2 * q3 * q4);
    J.R[6] = -(2 * q1 * q3 - 2 * q2 * q4); J.R[7] = 2 * q1 * q2 + 2 * q3 * q4; J.R[8] = 2 * q1 * q1 + 2 * q4 * q4 - 1;
    J.Jpq[0] = 4 * q1 * vx - 2 * q4 * vy + 2 * q3 * vz; J.Jpq[1] = 4 * q2 * vx + 2 * q3 * vy + 2 * q4 * vz;
    J.Jpq[2] = 2 * q2 * vy + 2 * q1 * vz;               J.Jpq[3] = -2 * q1 * vy + 2 * q2 * vz;
    J.Jpq[4] = 4 * q1 * vy + 2 * q4 * vx - 2 * q2 * vz; J.Jpq[5] = 2 * q3 * vx - 2 * q1 * vz;
    J.Jpq[6] = 4 * q3 * vy + 2 * q2 * vx + 2 * q4 * vz; J.Jpq[7] = 2 * q1 * vx + 2 * q3 * vz;
    J.Jpq[8] = 4 * q1 * vz - 2 * q3 * vx + 2 * q2 * vy; J.Jpq[9] = 2 * q4 * vx + 2 * q1 * vy;
    J.Jpq[10] = -2 * q1 * vx + 2 * q4 * vy;             J.Jpq[11] = 4 * q4 * vz + 2 * q2 * vx + 2 * q3 * vy;
}

// out = (df/dx)(point) * s  for a direction s whose q-part / w-part may be structurally zero.
// s, out: 13 entries (s[0..2] is never read: nothing depends on position).
template <bool HQ, bool HW>
__device__ __forceinline__ void jvp(const JacPoint& J, const double* __restrict__ s, double* __restrict__ o) {
    const double q1 = J.q[0], q2 = J.q[1], q3 = J.q[2], q4 = J.q[3];
    const double vx = J.v[0], vy = J.v[1], vz = J.v[2];
    const double wx = J.w[0], wy = J.w[1], wz = J.w[2];
    const double sv0 = s[7], sv1 = s[8], sv2 = s[9];
    double o0 = J.R[0] * sv0 + J.R[1] * sv1 + J.R[2] * sv2;
    double o1 = J.R[3] * sv0 + J.R[4] * sv1 + J.R[5] * sv2;
    double o2 = J.R[6] * sv0 + J.R[7] * sv1 + J.R[8] * sv2;
    double o3 = 0, o4 = 0, o5 = 0, o6 = 0;
    double o7 = wz * sv1 - wy * sv2;
    double o8 = -wz * sv0 + wx * sv2;
    double o9 = wy * sv0 - wx * sv1;
    double o10 = 0, o11 = 0, o12 = 0;
    if (HQ) {
        const double a = s[3], b = s[4], c = s[5], d = s[6];
        o0 += J.Jpq[0] * a + J.Jpq[1] * b + J.Jpq[2] * c + J.Jpq[3] * d;
        o1 += J.Jpq[4] * a + J.Jpq[5] * b + J.Jpq[6] * c + J.Jpq[7] * d;
        o2 += J.Jpq[8] * a + J.Jpq[9] * b + J.Jpq[10] * c + J.Jpq[11] * d;
        o3 += 0.5 * (-wx * b - wy * c - wz * d);
        o4 += 0.5 * (wx * a + wz * c - wy * d);
        o5 += 0.5 * (wy * a - wz * b + wx * d);
        o6 += 0.5 * (wz * a + wy * b - wx * c);
        o7 += 2 * G0 * (q3 * a - q4 * b + q1 * c - q2 * d);
        o8 += -2 * G0 * (q2 * a + q1 * b + q4 * c + q3 * d);
        o9 += -4 * G0 * (q1 * a + q4 * d);
    }
    if (HW) {
        const double a = s[10], b = s[11], c = s[12];
        o3 += 0.5 * (-q2 * a - q3 * b - q4 * c);
        o4 += 0.5 * (q1 * a - q4 * b + q3 * c);
        o5 += 0.5 * (q4 * a + q1 * b - q2 * c);
        o6 += 0.5 * (-q3 * a + q2 * b + q1 * c);
        o7 += -vz * b + vy * c;
        o8 += vz * a - vx * c;
        o9 += -vy * a + vx * b;
        o10 = KWX * (wz * b + wy * c);
        o11 = KWY * (wz * a + wx * c);
        o12 = KWZ * (wy * a + wx * b);
    }
    o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3; o[4] = o4; o[5] = o5; o[6] = o6;
    o[7] = o7; o[8] = o8; o[9] = o9; o[10] = o10; o[11] = o11; o[12] = o12;
}

// df/du column c (rows 9..12 only): d v'_z, d w'_x, d w'_y, d w'_z
__device__ __forceinline__ void ju_col(int c, const double* __restrict__ u, double* __restrict__ o4) {
    const double uc = 2.0 * u[c];
    const double sa = (c < 2) ? 1.0 : -1.0;             // w1 w2 | -w3 -w4
    const double sb = (c == 0 || c == 3) ? 1.0 : -1.0;  // w1 -w2 -w3 w4
    const double sc = (c == 0 || c == 2) ? 1.0 : -1.0;  // w1 -w2 w3 -w4
    o4[0] = KT * uc;
    o4[1] = KA * sa * uc;
    o4[2] = KB * sb * uc;
    o4[3] = KC * sc * uc;
}

// f(x, u) and (df/dx)(x) s + (df/du) du at one RK point, evaluated TOGETHER (export_ode_model.py:85-97 and its
// directional derivative, restated with shared sub-expressions: in this mapping a lane integrates one column only, so
// a Jacobian point kept for several columns -- cfnmpc_model.hpp: jac_point / jvp -- would be built for a single use).
//   xq[10] = q | v | w of the point, sq[10] = the direction's q | v | w parts (nothing depends on position);
//   rot[4] = the rotor terms of v_z', w' (functions of u only: constant over the interval), ju[4] = (df/du) du rows 9..12;
//   kk[13], dk[13]: slopes in EXTERNAL order.  With r = R(q) / 2:  p' = R v,  dp' = R dv + dR v.
__device__ __forceinline__ void lf_point(const double (&xq)[10], const double (&sq)[10], const double (&rot)[4],
                                         const double (&ju)[4], double (&kk)[13], double (&dk)[13]) {
    const double q1 = xq[0], q2 = xq[1], q3 = xq[2], q4 = xq[3], vx = xq[4], vy = xq[5], vz = xq[6];
    const double wx = xq[7], wy = xq[8], wz = xq[9];
    const double a = sq[0], b = sq[1], c = sq[2], d = sq[3], sx = sq[4], sy = sq[5], sz = sq[6];
    const double ox = sq[7], oy = sq[8], oz = sq[9];
    // r = R / 2
    const double t11 = q1 * q1 - 0.5;
    const double r0 = q2 * q2 + t11, r4 = q3 * q3 + t11, r8 = q4 * q4 + t11;
    const double p14 = q1 * q4, p13 = q1 * q3, p12 = q1 * q2;
    const double r1 = q2 * q3 - p14, r3 = q2 * q3 + p14;
    const double r2 = q2 * q4 + p13, r6 = q2 * q4 - p13;
    const double r5 = q3 * q4 - p12, r7 = q3 * q4 + p12;
    // dr = dR / 2 (the diagonal entries halved once more: they meet 2 v below)
    const double e1 = q1 * a;
    const double h0 = q2 * b + e1, h4 = q3 * c + e1, h8 = q4 * d + e1;
    const double d23 = q2 * c + q3 * b, d14 = q1 * d + q4 * a;
    const double d24 = q2 * d + q4 * b, d13 = q1 * c + q3 * a;
    const double d34 = q3 * d + q4 * c, d12 = q1 * b + q2 * a;
    const double dr1 = d23 - d14, dr3 = d23 + d14;
    const double dr2 = d24 + d13, dr6 = d24 - d13;
    const double dr5 = d34 - d12, dr7 = d34 + d12;
    const double v2x = 2.0 * vx, v2y = 2.0 * vy, v2z = 2.0 * vz;
    // position rows
    kk[0] = r0 * v2x + r1 * v2y + r2 * v2z;
    kk[1] = r3 * v2x + r4 * v2y + r5 * v2z;
    kk[2] = r6 * v2x + r7 * v2y + r8 * v2z;
    dk[0] = 2.0 * (r0 * sx + r1 * sy + r2 * sz + h0 * v2x + dr1 * vy + dr2 * vz);
    dk[1] = 2.0 * (r3 * sx + r4 * sy + r5 * sz + dr3 * vx + h4 * v2y + dr5 * vz);
    dk[2] = 2.0 * (r6 * sx + r7 * sy + r8 * sz + dr6 * vx + dr7 * vy + h8 * v2z);
    // quaternion rows (halved rates)
    const double hx = 0.5 * wx, hy = 0.5 * wy, hz = 0.5 * wz;
    const double gx = 0.5 * ox, gy = 0.5 * oy, gz = 0.5 * oz;
    kk[3] = -(q2 * hx + q3 * hy + q4 * hz);
    kk[4] = q1 * hx - q4 * hy + q3 * hz;
    kk[5] = q4 * hx + q1 * hy - q2 * hz;
    kk[6] = q2 * hy - q3 * hx + q1 * hz;
    dk[3] = -(b * hx + c * hy + d * hz + q2 * gx + q3 * gy + q4 * gz);
    dk[4] = a * hx - d * hy + c * hz + q1 * gx - q4 * gy + q3 * gz;
    dk[5] = d * hx + a * hy - b * hz + q4 * gx + q1 * gy - q2 * gz;
    dk[6] = b * hy - c * hx + a * hz + q2 * gy - q3 * gx + q1 * gz;
    // body velocity rows: v' = v x w - g0 R' e_z (+ thrust);  -G0 R[6..8] = -2 G0 r[6..8]
    kk[7] = vy * wz - vz * wy - (2.0 * G0) * r6;
    kk[8] = vz * wx - vx * wz - (2.0 * G0) * r7;
    kk[9] = vx * wy - vy * wx - (2.0 * G0) * r8 + rot[0];
    dk[7] = sy * wz + vy * oz - sz * wy - vz * oy - (2.0 * G0) * dr6;
    dk[8] = sz * wx + vz * ox - sx * wz - vx * oz - (2.0 * G0) * dr7;
    dk[9] = sx * wy + vx * oy - sy * wx - vy * ox - (4.0 * G0) * h8 + ju[0];
    // body rate rows
    kk[10] = KWX * (wy * wz) + rot[1];
    kk[11] = KWY * (wx * wz) + rot[2];
    kk[12] = KWZ * (wx * wy) + rot[3];
    dk[10] = KWX * (oy * wz + wy * oz) + ju[1];
    dk[11] = KWY * (ox * wz + wx * oz) + ju[2];
    dk[12] = KWZ * (ox * wy + wx * oy) + ju[3];
}

}  // namespace cfn
