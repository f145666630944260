// cfnmpc_kernels.hip -- HIP kernels of the batched Crazyflie SQP-RTI step (gfx950, FP64).
//
// v1 mapping: ONE NMPC INSTANCE PER WAVEFRONT LANE.  All per-instance data live in HBM in
// SoA form with the instance index fastest:
//     field[(stage * E + elem) * Bp + inst]           (Bp = batch padded to 64)
// so that the 64 lanes of a wave read/write 512 contiguous bytes per element.  Instances are
// independent, so there is no inter-lane or inter-wave communication at all; each wave runs
// its own interior-point loop until all of its lanes have converged (wave-uniform exit via
// __any), converged lanes are predicated off.
//
// Kernels:
//   k_linearise : RK4 + forward sensitivities per shooting interval (acados sim_erk + CasADi
//                 forw_vde, acados_mpc.cpp:84), column-wise so that only one 13-vector of the
//                 sensitivity is live; writes A (compact 97), B (52), b (13) per stage.
//   k_qp_ipm    : box-constrained OCP-QP by Mehrotra predictor-corrector with stage-wise
//                 Riccati sweeps (HPIPM's role, generate_c_code.py:140), delta form; then the
//                 full RTI step (iterate += step) and per-instance statistics.
//   k_sim       : RK4 predictor / plant step (acados_estimator.cpp:573-593).
//   k_aos2soa / k_soa2aos / k_init_iterate : layout glue for the C-ABI.
#include <hip/hip_runtime.h>

#include "cfnmpc_model.hpp"
#include "cfnmpc_ws.hpp"

namespace cfn {

#define IDX(k, e, E) (((size_t)(k) * (E) + (e)) * (size_t)P.Bp + (size_t)inst)

// =============================================================================================
// linearisation
// =============================================================================================
template <bool HQ, bool HW, bool IS_U>
__device__ __forceinline__ void sens_column(const JacPoint (&J)[4], const double* __restrict__ u, int c,
                                            double h, double* __restrict__ col) {
    // c: state column (3..12) or input column (0..3) when IS_U
    double s0[13], s[13], k1[13], k2[13], k3[13], k4[13], ju[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 13; i++) s0[i] = 0.0;
    if (!IS_U) s0[c] = 1.0; else ju_col(c, u, ju);
    jvp<HQ, HW>(J[0], s0, k1);
#pragma unroll
    for (int i = 0; i < 4; i++) k1[9 + i] += ju[i];
#pragma unroll
    for (int i = 0; i < 13; i++) s[i] = s0[i] + 0.5 * h * k1[i];
    jvp<HQ, HW>(J[1], s, k2);
#pragma unroll
    for (int i = 0; i < 4; i++) k2[9 + i] += ju[i];
#pragma unroll
    for (int i = 0; i < 13; i++) s[i] = s0[i] + 0.5 * h * k2[i];
    jvp<HQ, HW>(J[2], s, k3);
#pragma unroll
    for (int i = 0; i < 4; i++) k3[9 + i] += ju[i];
#pragma unroll
    for (int i = 0; i < 13; i++) s[i] = s0[i] + h * k3[i];
    jvp<HQ, HW>(J[3], s, k4);
#pragma unroll
    for (int i = 0; i < 4; i++) k4[9 + i] += ju[i];
#pragma unroll
    for (int i = 0; i < 13; i++) col[i] = s0[i] + (h / 6.0) * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
}

__global__ __launch_bounds__(64) void k_linearise(Params P) {
    const int inst = blockIdx.x * 64 + threadIdx.x;
    if (inst >= P.B) return;
    const double h = P.dt;
    double xn[13];
#pragma unroll
    for (int e = 0; e < 13; e++) xn[e] = P.xit[IDX(0, e, 13)];
    for (int k = 0; k < P.N; k++) {
        double x[13], u[4], xt[13], k1[13], k2[13], k3[13], k4[13];
        JacPoint J[4];
#pragma unroll
        for (int e = 0; e < 13; e++) x[e] = xn[e];
#pragma unroll
        for (int e = 0; e < 4; e++) u[e] = P.uit[IDX(k, e, 4)];
#pragma unroll
        for (int e = 0; e < 13; e++) xn[e] = P.xit[IDX(k + 1, e, 13)];
        // nominal RK4 (classic tableau, one step per interval)
        f_expl(x, u, k1);
        jac_point(x, J[0]);
#pragma unroll
        for (int e = 0; e < 13; e++) xt[e] = x[e] + 0.5 * h * k1[e];
        f_expl(xt, u, k2);
        jac_point(xt, J[1]);
#pragma unroll
        for (int e = 0; e < 13; e++) xt[e] = x[e] + 0.5 * h * k2[e];
        f_expl(xt, u, k3);
        jac_point(xt, J[2]);
#pragma unroll
        for (int e = 0; e < 13; e++) xt[e] = x[e] + h * k3[e];
        f_expl(xt, u, k4);
        jac_point(xt, J[3]);
#pragma unroll
        for (int e = 0; e < 13; e++) {
            const double phi = x[e] + (h / 6.0) * (k1[e] + 2 * k2[e] + 2 * k3[e] + k4[e]);
            P.b[IDX(k, e, 13)] = phi - xn[e];
        }
        // sensitivities, one column at a time
        double col[13];
#pragma unroll
        for (int c = 3; c < 7; c++) {  // quaternion columns: rows p,q,v
            sens_column<true, false, false>(J, u, c, h, col);
#pragma unroll
            for (int r = 0; r < 10; r++) P.A[IDX(k, a_idx(r, c), A_NNZ)] = col[r];
        }
#pragma unroll
        for (int c = 7; c < 10; c++) {  // velocity columns: rows p,v
            sens_column<false, false, false>(J, u, c, h, col);
#pragma unroll
            for (int r = 0; r < 3; r++) P.A[IDX(k, a_idx(r, c), A_NNZ)] = col[r];
#pragma unroll
            for (int r = 7; r < 10; r++) P.A[IDX(k, a_idx(r, c), A_NNZ)] = col[r];
        }
#pragma unroll
        for (int c = 10; c < 13; c++) {  // rate columns: all rows
            sens_column<true, true, false>(J, u, c, h, col);
#pragma unroll
            for (int r = 0; r < 13; r++) P.A[IDX(k, a_idx(r, c), A_NNZ)] = col[r];
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {  // input columns: all rows
            sens_column<true, true, true>(J, u, c, h, col);
#pragma unroll
            for (int r = 0; r < 13; r++) P.Bm[IDX(k, r * 4 + c, 52)] = col[r];
        }
    }
}

// =============================================================================================
// Riccati sweeps (per lane)
// =============================================================================================
struct LaneIPM {
    double mu, res, alpha;
    int iters, status;
    bool act;
};

// symmetric positive definite 4x4 (packed upper) -> inverse (packed upper).  false if not SPD.
__device__ __forceinline__ bool spd4_inv(const double* __restrict__ S, double* __restrict__ Si) {
    double L[4][4], Li[4][4];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        double s = S[s4(j, j)];
#pragma unroll
        for (int k = 0; k < j; k++) s -= L[j][k] * L[j][k];
        ok = ok && (s > 0.0);
        const double ljj = sqrt(s), inv = 1.0 / ljj;
        L[j][j] = ljj;
        Li[j][j] = inv;
#pragma unroll
        for (int i = j + 1; i < 4; i++) {
            double t = S[s4(i, j)];
#pragma unroll
            for (int k = 0; k < j; k++) t -= L[i][k] * L[j][k];
            L[i][j] = t * inv;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = j + 1; i < 4; i++) {
            double t = 0;
#pragma unroll
            for (int k = j; k < i; k++) t -= L[i][k] * Li[k][j];
            Li[i][j] = t * Li[i][i];
        }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = i; j < 4; j++) {
            double t = 0;
#pragma unroll
            for (int k = j; k < 4; k++) t += Li[k][i] * Li[k][j];
            Si[s4(i, j)] = t;
        }
    return ok;
}

// Backward factorisation sweep.  ABSOLUTE: start solve with the QP's affine terms (q, b, r);
// otherwise homogeneous Newton system with input Hessian R + Dl + Du and gradient g_aff that
// are formed on the fly from the interior-point state.  Writes K, Sinv, d per stage.
template <bool ABSOLUTE>
__device__ __forceinline__ bool sweep_factor(const Params& P, const int inst) {
    double Pm[S_NNZ], p[13];
    bool ok = true;
#pragma unroll
    for (int i = 0; i < S_NNZ; i++) Pm[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 13; i++) {
        Pm[sidx(i, i)] = P.WN[i];
        p[i] = ABSOLUTE ? P.WN[i] * (P.xit[IDX(P.N, i, 13)] - P.yref_e[IDX(0, i, 13)]) : 0.0;
    }
    for (int k = P.N - 1; k >= 0; k--) {
        double a[A_NNZ], bm[52];
#pragma unroll
        for (int e = 0; e < A_NNZ; e++) a[e] = P.A[IDX(k, e, A_NNZ)];
#pragma unroll
        for (int e = 0; e < 52; e++) bm[e] = P.Bm[IDX(k, e, 52)];
        // input Hessian / gradient of this stage
        double Rh[4], g[4], uk[4];
#pragma unroll
        for (int e = 0; e < 4; e++) uk[e] = P.uit[IDX(k, e, 4)];
        if (ABSOLUTE) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                Rh[e] = P.W[13 + e];
                g[e] = P.W[13 + e] * (uk[e] - P.yref[IDX(k, 13 + e, 17)]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const double v = P.v[IDX(k, e, 4)], tl = P.tl[IDX(k, e, 4)], tu = P.tu[IDX(k, e, 4)];
                const double ll = P.ll[IDX(k, e, 4)], lu = P.lu[IDX(k, e, 4)], rg = P.rg[IDX(k, e, 4)];
                const double lb = P.u_min - uk[e], ub = P.u_max - uk[e];
                const double rl = v - lb - tl, ru = ub - v - tu;
                const double Dl = ll / tl, Du = lu / tu;
                Rh[e] = P.W[13 + e] + Dl + Du;
                g[e] = rg + ll + Dl * rl - lu - Du * ru;
            }
        }
        // hb = p + P b (absolute only)
        double hb[13];
        if (ABSOLUTE) {
            double bv[13];
#pragma unroll
            for (int e = 0; e < 13; e++) bv[e] = P.b[IDX(k, e, 13)];
#pragma unroll
            for (int i = 0; i < 13; i++) {
                double s = p[i];
#pragma unroll
                for (int l = 0; l < 13; l++) s += Pm[sidx(i, l)] * bv[l];
                hb[i] = s;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 13; i++) hb[i] = p[i];
        }
        // PB = P B (13x4), S = Rh + B'PB (sym 4x4), rho = g + B'hb
        double PB[52], S[10], rho[4];
#pragma unroll
        for (int i = 0; i < 13; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                double s = 0;
#pragma unroll
                for (int l = 0; l < 13; l++) s += Pm[sidx(i, l)] * bm[l * 4 + j];
                PB[i * 4 + j] = s;
            }
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int j = i; j < 4; j++) {
                double s = (i == j) ? Rh[i] : 0.0;
#pragma unroll
                for (int l = 0; l < 13; l++) s += bm[l * 4 + i] * PB[l * 4 + j];
                S[s4(i, j)] = s;
            }
            double s = g[i];
#pragma unroll
            for (int l = 0; l < 13; l++) s += bm[l * 4 + i] * hb[l];
            rho[i] = s;
        }
        double Si[10];
        ok = spd4_inv(S, Si) && ok;
        // column-wise: w_j = P a_j ; G[:,j] = B' w_j ; M[i][j] = a_i . w_j (i <= j)
        double G[52], Pn[S_NNZ];
#pragma unroll
        for (int j = 0; j < 13; j++) {
            double w[13];
#pragma unroll
            for (int i = 0; i < 13; i++) {
                double s = 0;
#pragma unroll
                for (int l = 0; l < 13; l++) {
                    if (a_kind(l, j) == 1) s += Pm[sidx(i, l)];
                    if (a_kind(l, j) == 2) s += Pm[sidx(i, l)] * a[a_idx(l, j)];
                }
                w[i] = s;
            }
#pragma unroll
            for (int c = 0; c < 4; c++) {
                double s = 0;
#pragma unroll
                for (int l = 0; l < 13; l++) s += bm[l * 4 + c] * w[l];
                G[c * 13 + j] = s;
            }
#pragma unroll
            for (int i = 0; i <= j; i++) {
                double s = (i == j) ? P.W[i] : 0.0;
#pragma unroll
                for (int l = 0; l < 13; l++) {
                    if (a_kind(l, i) == 1) s += w[l];
                    if (a_kind(l, i) == 2) s += a[a_idx(l, i)] * w[l];
                }
                Pn[sidx(i, j)] = s;
            }
        }
        // K = Sinv G, d = Sinv rho
        double K[52], d[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
#pragma unroll
            for (int j = 0; j < 13; j++) {
                double s = 0;
#pragma unroll
                for (int l = 0; l < 4; l++) s += Si[s4(c, l)] * G[l * 13 + j];
                K[c * 13 + j] = s;
            }
            double s = 0;
#pragma unroll
            for (int l = 0; l < 4; l++) s += Si[s4(c, l)] * rho[l];
            d[c] = s;
        }
        // P <- Q + A'PA - G'K ; p <- (q +) A'hb - K'rho
#pragma unroll
        for (int i = 0; i < 13; i++)
#pragma unroll
            for (int j = i; j < 13; j++) {
                double s = Pn[sidx(i, j)];
#pragma unroll
                for (int l = 0; l < 4; l++) s -= G[l * 13 + i] * K[l * 13 + j];
                Pm[sidx(i, j)] = s;
            }
#pragma unroll
        for (int i = 0; i < 13; i++) {
            double s = ABSOLUTE ? P.W[i] * (P.xit[IDX(k, i, 13)] - P.yref[IDX(k, i, 17)]) : 0.0;
#pragma unroll
            for (int l = 0; l < 13; l++) {
                if (a_kind(l, i) == 1) s += hb[l];
                if (a_kind(l, i) == 2) s += a[a_idx(l, i)] * hb[l];
            }
#pragma unroll
            for (int l = 0; l < 4; l++) s -= K[l * 13 + i] * rho[l];
            p[i] = s;
        }
#pragma unroll
        for (int e = 0; e < 52; e++) P.K[IDX(k, e, 52)] = K[e];
#pragma unroll
        for (int e = 0; e < 10; e++) P.Sinv[IDX(k, e, 10)] = Si[e];
#pragma unroll
        for (int e = 0; e < 4; e++) P.d[IDX(k, e, 4)] = d[e];
    }
    return ok;
}

// x+ = A x + B v (+ b) with the compact A
template <bool WITH_B>
__device__ __forceinline__ void propagate(const Params& P, const int inst, const int k, double* __restrict__ x,
                                          const double* __restrict__ v) {
    double xn[13];
#pragma unroll
    for (int i = 0; i < 13; i++) xn[i] = WITH_B ? P.b[IDX(k, i, 13)] : 0.0;
#pragma unroll
    for (int i = 0; i < 13; i++)
#pragma unroll
        for (int l = 0; l < 13; l++) {
            if (a_kind(i, l) == 1) xn[i] += x[l];
            if (a_kind(i, l) == 2) xn[i] += P.A[IDX(k, a_idx(i, l), A_NNZ)] * x[l];
        }
#pragma unroll
    for (int i = 0; i < 13; i++)
#pragma unroll
        for (int c = 0; c < 4; c++) xn[i] += P.Bm[IDX(k, i * 4 + c, 52)] * v[c];
#pragma unroll
    for (int i = 0; i < 13; i++) x[i] = xn[i];
}

__device__ __forceinline__ void feedback(const Params& P, const int inst, const int k, const double* __restrict__ x,
                                         double* __restrict__ v) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
        double s = -P.d[IDX(k, c, 4)];
#pragma unroll
        for (int l = 0; l < 13; l++) s -= P.K[IDX(k, c * 13 + l, 52)] * x[l];
        v[c] = s;
    }
}

__device__ __forceinline__ double ratio(double z, double dz, double a) {
    const double t = -z / dz;
    return (dz < 0.0 && t < a) ? t : a;
}

// Backward sweep re-using the factorisation for the corrector right-hand side; overwrites d.
__device__ __forceinline__ void sweep_resolve(const Params& P, const int inst, const double smu) {
    double p[13];
#pragma unroll
    for (int i = 0; i < 13; i++) p[i] = 0.0;
    for (int k = P.N - 1; k >= 0; k--) {
        double rho[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const double uk = P.uit[IDX(k, e, 4)];
            const double v = P.v[IDX(k, e, 4)], tl = P.tl[IDX(k, e, 4)], tu = P.tu[IDX(k, e, 4)];
            const double ll = P.ll[IDX(k, e, 4)], lu = P.lu[IDX(k, e, 4)];
            const double dva = P.dva[IDX(k, e, 4)];
            const double lb = P.u_min - uk, ub = P.u_max - uk;
            const double rl = v - lb - tl, ru = ub - v - tu;
            const double dtl = dva + rl, dtu = -dva + ru;
            const double dll = -ll - (ll / tl) * dtl, dlu = -lu - (lu / tu) * dtu;
            const double cl = dll * dtl, cu = dlu * dtu;
            double s = (cl - smu) / tl - (cu - smu) / tu;
#pragma unroll
            for (int l = 0; l < 13; l++) s += P.Bm[IDX(k, l * 4 + e, 52)] * p[l];
            rho[e] = s;
        }
        double Si[10];
#pragma unroll
        for (int e = 0; e < 10; e++) Si[e] = P.Sinv[IDX(k, e, 10)];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            double s = 0;
#pragma unroll
            for (int l = 0; l < 4; l++) s += Si[s4(c, l)] * rho[l];
            P.d[IDX(k, c, 4)] = s;
        }
        double pn[13];
#pragma unroll
        for (int i = 0; i < 13; i++) {
            double s = 0;
#pragma unroll
            for (int l = 0; l < 13; l++) {
                if (a_kind(l, i) == 1) s += p[l];
                if (a_kind(l, i) == 2) s += P.A[IDX(k, a_idx(l, i), A_NNZ)] * p[l];
            }
#pragma unroll
            for (int l = 0; l < 4; l++) s -= P.K[IDX(k, l * 13 + i, 52)] * rho[l];
            pn[i] = s;
        }
#pragma unroll
        for (int i = 0; i < 13; i++) p[i] = pn[i];
    }
}

// =============================================================================================
// QP solve + RTI update
// =============================================================================================
__global__ __launch_bounds__(64) void k_qp_ipm(Params P) {
    const int inst = blockIdx.x * 64 + threadIdx.x;
    const bool valid = inst < P.B;
    LaneIPM L;
    L.iters = 0;
    L.status = 0;
    L.res = 0.0;
    L.act = false;
    const double nc = 8.0 * P.N;

    if (valid) {
        // ---- start: unconstrained minimiser (absolute Riccati solve)
        bool ok = sweep_factor<true>(P, inst);
        double x[13];
#pragma unroll
        for (int e = 0; e < 13; e++) x[e] = P.x0[IDX(0, e, 13)] - P.xit[IDX(0, e, 13)];
        bool feas = true;
        double viol = 0.0;
        for (int k = 0; k < P.N; k++) {
            double v[4];
            feedback(P, inst, k, x, v);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const double uk = P.uit[IDX(k, e, 4)];
                const double lb = P.u_min - uk, ub = P.u_max - uk;
                feas = feas && (v[e] >= lb) && (v[e] <= ub);
                viol = fmax(viol, fmax(lb - v[e], v[e] - ub));
                P.v[IDX(k, e, 4)] = v[e];
            }
            propagate<true>(P, inst, k, x, v);
        }
        if (!ok || !(viol == viol)) {
            L.status = 4;
            L.res = nan("");
        } else if (!feas) {
            // ---- shift slacks / multipliers positive; residuals of the start
            const double mu0 = fmax(viol, P.lam0_min);
            double mu = 0.0, res = 0.0;
            for (int k = 0; k < P.N; k++) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const double uk = P.uit[IDX(k, e, 4)], v = P.v[IDX(k, e, 4)];
                    const double lb = P.u_min - uk, ub = P.u_max - uk;
                    const double tl = fmax(v - lb, P.thr0), tu = fmax(ub - v, P.thr0);
                    const double ll = mu0 / tl, lu = mu0 / tu, rg = -ll + lu;
                    P.tl[IDX(k, e, 4)] = tl; P.tu[IDX(k, e, 4)] = tu;
                    P.ll[IDX(k, e, 4)] = ll; P.lu[IDX(k, e, 4)] = lu;
                    P.rg[IDX(k, e, 4)] = rg;
                    mu += ll * tl + lu * tu;
                    res = fmax(res, fmax(ll * tl, lu * tu));
                    res = fmax(res, fmax(fabs(rg), fmax(fabs(v - lb - tl), fabs(ub - v - tu))));
                }
            }
            L.mu = mu / nc;
            L.res = res;
            L.act = true;
        }
    }

    // ---- interior-point loop, wave-uniform trip count
    while (__any(L.act)) {
        if (L.act) {
            if (!(L.res == L.res)) { L.status = 4; L.act = false; }
            else if (L.res <= P.tol) { L.status = 0; L.act = false; }
            else if (L.iters >= P.max_iter) { L.status = 2; L.act = false; }
        }
        if (!__any(L.act)) break;
        if (L.act) {
            L.iters++;
            // predictor: factorise, forward
            const bool ok = sweep_factor<false>(P, inst);
            double x[13];
#pragma unroll
            for (int e = 0; e < 13; e++) x[e] = 0.0;
            double a = 1.0;
            for (int k = 0; k < P.N; k++) {
                double dv[4];
                feedback(P, inst, k, x, dv);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const double uk = P.uit[IDX(k, e, 4)];
                    const double v = P.v[IDX(k, e, 4)], tl = P.tl[IDX(k, e, 4)], tu = P.tu[IDX(k, e, 4)];
                    const double ll = P.ll[IDX(k, e, 4)], lu = P.lu[IDX(k, e, 4)];
                    const double lb = P.u_min - uk, ub = P.u_max - uk;
                    const double rl = v - lb - tl, ru = ub - v - tu;
                    const double dtl = dv[e] + rl, dtu = -dv[e] + ru;
                    const double dll = -ll - (ll / tl) * dtl, dlu = -lu - (lu / tu) * dtu;
                    a = ratio(tl, dtl, a); a = ratio(tu, dtu, a);
                    a = ratio(ll, dll, a); a = ratio(lu, dlu, a);
                    P.dva[IDX(k, e, 4)] = dv[e];
                }
                propagate<false>(P, inst, k, x, dv);
            }
            // mu_aff
            double mu_aff = 0.0;
            for (int k = 0; k < P.N; k++) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const double uk = P.uit[IDX(k, e, 4)];
                    const double v = P.v[IDX(k, e, 4)], tl = P.tl[IDX(k, e, 4)], tu = P.tu[IDX(k, e, 4)];
                    const double ll = P.ll[IDX(k, e, 4)], lu = P.lu[IDX(k, e, 4)];
                    const double dva = P.dva[IDX(k, e, 4)];
                    const double lb = P.u_min - uk, ub = P.u_max - uk;
                    const double rl = v - lb - tl, ru = ub - v - tu;
                    const double dtl = dva + rl, dtu = -dva + ru;
                    const double dll = -ll - (ll / tl) * dtl, dlu = -lu - (lu / tu) * dtu;
                    mu_aff += (ll + a * dll) * (tl + a * dtl) + (lu + a * dlu) * (tu + a * dtu);
                }
            }
            mu_aff /= nc;
            const double sr = mu_aff / L.mu, smu = sr * sr * sr * L.mu;
            // corrector: re-solve, forward
            sweep_resolve(P, inst, smu);
#pragma unroll
            for (int e = 0; e < 13; e++) x[e] = 0.0;
            a = 1.0;
            for (int k = 0; k < P.N; k++) {
                double dvc[4];
                feedback(P, inst, k, x, dvc);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const double uk = P.uit[IDX(k, e, 4)];
                    const double v = P.v[IDX(k, e, 4)], tl = P.tl[IDX(k, e, 4)], tu = P.tu[IDX(k, e, 4)];
                    const double ll = P.ll[IDX(k, e, 4)], lu = P.lu[IDX(k, e, 4)];
                    const double dva = P.dva[IDX(k, e, 4)];
                    const double lb = P.u_min - uk, ub = P.u_max - uk;
                    const double rl = v - lb - tl, ru = ub - v - tu;
                    const double dtla = dva + rl, dtua = -dva + ru;
                    const double Dl = ll / tl, Du = lu / tu;
                    const double cl = (-ll - Dl * dtla) * dtla, cu = (-lu - Du * dtua) * dtua;
                    const double dv = dva + dvc[e];
                    const double dtl = dv + rl, dtu = -dv + ru;
                    const double dll = (smu - cl) / tl - ll - Dl * dtl, dlu = (smu - cu) / tu - lu - Du * dtu;
                    a = ratio(tl, dtl, a); a = ratio(tu, dtu, a);
                    a = ratio(ll, dll, a); a = ratio(lu, dlu, a);
                    P.dvc[IDX(k, e, 4)] = dvc[e];
                }
                propagate<false>(P, inst, k, x, dvc);
            }
            a = fmin(1.0, P.tau * a);
            // update + residuals of the new point
            double mu = 0.0, res = 0.0;
            for (int k = 0; k < P.N; k++) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const double uk = P.uit[IDX(k, e, 4)];
                    double v = P.v[IDX(k, e, 4)], tl = P.tl[IDX(k, e, 4)], tu = P.tu[IDX(k, e, 4)];
                    double ll = P.ll[IDX(k, e, 4)], lu = P.lu[IDX(k, e, 4)], rg = P.rg[IDX(k, e, 4)];
                    const double dva = P.dva[IDX(k, e, 4)], dvc = P.dvc[IDX(k, e, 4)];
                    const double lb = P.u_min - uk, ub = P.u_max - uk;
                    const double rl = v - lb - tl, ru = ub - v - tu;
                    const double dtla = dva + rl, dtua = -dva + ru;
                    const double Dl = ll / tl, Du = lu / tu;
                    const double cl = (-ll - Dl * dtla) * dtla, cu = (-lu - Du * dtua) * dtua;
                    const double dv = dva + dvc;
                    const double dtl = dv + rl, dtu = -dv + ru;
                    const double dll = (smu - cl) / tl - ll - Dl * dtl, dlu = (smu - cu) / tu - lu - Du * dtu;
                    v += a * dv; tl += a * dtl; tu += a * dtu; ll += a * dll; lu += a * dlu;
                    rg *= (1.0 - a);
                    P.v[IDX(k, e, 4)] = v; P.tl[IDX(k, e, 4)] = tl; P.tu[IDX(k, e, 4)] = tu;
                    P.ll[IDX(k, e, 4)] = ll; P.lu[IDX(k, e, 4)] = lu; P.rg[IDX(k, e, 4)] = rg;
                    mu += ll * tl + lu * tu;
                    res = fmax(res, fmax(ll * tl, lu * tu));
                    res = fmax(res, fmax(fabs(rg), fmax(fabs(v - lb - tl), fabs(ub - v - tu))));
                }
            }
            L.mu = mu / nc;
            L.res = ok ? res : nan("");
        }
    }

    // ---- expand (dynamics-exact state roll-out of the final inputs) + full RTI step
    if (valid) {
        double x[13];
#pragma unroll
        for (int e = 0; e < 13; e++) x[e] = P.x0[IDX(0, e, 13)] - P.xit[IDX(0, e, 13)];
        if (L.status != 4) {
            for (int k = 0; k < P.N; k++) {
                double v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = P.v[IDX(k, e, 4)];
                double xk[13];
#pragma unroll
                for (int e = 0; e < 13; e++) xk[e] = x[e];
                propagate<true>(P, inst, k, x, v);
#pragma unroll
                for (int e = 0; e < 13; e++) P.xit[IDX(k, e, 13)] += xk[e];
#pragma unroll
                for (int e = 0; e < 4; e++) P.uit[IDX(k, e, 4)] += v[e];
            }
#pragma unroll
            for (int e = 0; e < 13; e++) P.xit[IDX(P.N, e, 13)] += x[e];
        }
        P.status[inst] = L.status;
        P.iters[inst] = L.iters;
        P.res[inst] = L.res;
    }
}

// =============================================================================================
// predictor / plant step, layout glue
// =============================================================================================
__global__ void k_sim(int B, const double* __restrict__ x, const double* __restrict__ u, double T, int steps,
                      double* __restrict__ xn) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    double xc[13], uc[4], k1[13], k2[13], k3[13], k4[13], xt[13];
#pragma unroll
    for (int e = 0; e < 13; e++) xc[e] = x[(size_t)i * 13 + e];
#pragma unroll
    for (int e = 0; e < 4; e++) uc[e] = u[(size_t)i * 4 + e];
    const double h = T / steps;
    for (int s = 0; s < steps; s++) {
        f_expl(xc, uc, k1);
#pragma unroll
        for (int e = 0; e < 13; e++) xt[e] = xc[e] + 0.5 * h * k1[e];
        f_expl(xt, uc, k2);
#pragma unroll
        for (int e = 0; e < 13; e++) xt[e] = xc[e] + 0.5 * h * k2[e];
        f_expl(xt, uc, k3);
#pragma unroll
        for (int e = 0; e < 13; e++) xt[e] = xc[e] + h * k3[e];
        f_expl(xt, uc, k4);
#pragma unroll
        for (int e = 0; e < 13; e++) xc[e] += (h / 6.0) * (k1[e] + 2 * k2[e] + 2 * k3[e] + k4[e]);
    }
#pragma unroll
    for (int e = 0; e < 13; e++) xn[(size_t)i * 13 + e] = xc[e];
}

// AoS [B][S][E] (caller) -> SoA [S][E][Bp] (workspace) and back
__global__ void k_aos2soa(int B, int Bp, int S, int E, const double* __restrict__ aos, double* __restrict__ soa) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const int SE = S * E;
    for (int se = 0; se < SE; se++) soa[(size_t)se * Bp + i] = aos[(size_t)i * SE + se];
}
__global__ void k_soa2aos(int B, int Bp, int S, int E, int s0, int Stot, const double* __restrict__ soa,
                          double* __restrict__ aos) {
    // copies stages s0 .. s0+S-1 of an SoA field with Stot stages into AoS [B][S][E]
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    for (int s = 0; s < S; s++)
        for (int e = 0; e < E; e++) aos[((size_t)i * S + s) * E + e] = soa[((size_t)(s0 + s) * E + e) * Bp + i];
}
__global__ void k_init_iterate(Params P, int mode) {
    const int inst = blockIdx.x * blockDim.x + threadIdx.x;
    if (inst >= P.B) return;
    // generate_c_code.py:58,135 / SURVEY App. D-3
    const double hov = sqrt((MQ * G0) / (4 * CT));
    for (int k = 0; k <= P.N; k++)
        for (int e = 0; e < 13; e++)
            P.xit[IDX(k, e, 13)] = (mode == 1) ? P.x0[IDX(0, e, 13)] : (e == 3 ? 1.0 : 0.0);
    for (int k = 0; k < P.N; k++)
        for (int e = 0; e < 4; e++) P.uit[IDX(k, e, 4)] = (mode == 1) ? hov : 0.0;
}

}  // namespace cfn

// ---------------------------------------------------------------------------------------------
// launchers (called from cfnmpc_api.cpp through plain C++ declarations in cfnmpc_ws.hpp)
// ---------------------------------------------------------------------------------------------
namespace cfn {

void launch_linearise(const Params& P, hipStream_t st) {
    hipLaunchKernelGGL(k_linearise, dim3((P.B + 63) / 64), dim3(64), 0, st, P);
}
void launch_qp(const Params& P, hipStream_t st) {
    hipLaunchKernelGGL(k_qp_ipm, dim3((P.B + 63) / 64), dim3(64), 0, st, P);
}
void launch_sim(int B, const double* x, const double* u, double T, int steps, double* xn, hipStream_t st) {
    hipLaunchKernelGGL(k_sim, dim3((B + 255) / 256), dim3(256), 0, st, B, x, u, T, steps, xn);
}
void launch_aos2soa(int B, int Bp, int S, int E, const double* aos, double* soa, hipStream_t st) {
    hipLaunchKernelGGL(k_aos2soa, dim3((B + 255) / 256), dim3(256), 0, st, B, Bp, S, E, aos, soa);
}
void launch_soa2aos(int B, int Bp, int S, int E, int s0, int Stot, const double* soa, double* aos, hipStream_t st) {
    hipLaunchKernelGGL(k_soa2aos, dim3((B + 255) / 256), dim3(256), 0, st, B, Bp, S, E, s0, Stot, soa, aos);
}
void launch_init_iterate(const Params& P, int mode, hipStream_t st) {
    hipLaunchKernelGGL(k_init_iterate, dim3((P.B + 255) / 256), dim3(256), 0, st, P, mode);
}

}  // namespace cfn
