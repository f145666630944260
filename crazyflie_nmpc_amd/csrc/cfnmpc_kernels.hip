// cfnmpc_kernels.hip -- HIP kernels of the batched Crazyflie SQP-RTI step (gfx950, FP64).
//
// Mapping: one NMPC instance per 16-lane DPP row, four instances per wavefront, one wavefront
// per workgroup (cfnmpc_ws.hpp).  Instances never communicate; every wave runs its own
// interior-point loop until its four instances are done (wave-uniform trip count via __any).
//
// Kernels
//   k_linearise : RK4 + forward sensitivities per shooting interval (the role of acados
//                 sim_erk + CasADi forw_vde, acados_mpc.cpp:84): lane c integrates sensitivity
//                 COLUMN c; a 13x17 LDS tile re-distributes it into the row / column forms.
//   k_qp        : box-constrained OCP-QP by Mehrotra predictor-corrector over stage-wise
//                 Riccati sweeps in delta form (HPIPM's role, generate_c_code.py:140), then
//                 expansion and the full RTI step (acados_solve(), acados_mpc.cpp:611).
//   k_sim       : RK4 predictor / plant step (acados_estimator.cpp:573-593).
//   k_put / k_get / k_init_iterate : layout glue for the C-ABI.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "cfnmpc_dpp.hpp"
#include "cfnmpc_model.hpp"
#include "cfnmpc_ws.hpp"

namespace cfn {

// ---------------------------------------------------------------------------------------------
// small tools
// ---------------------------------------------------------------------------------------------
template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}
#define SFOR(var, lo, hi, ...) sfor<lo, hi>([&](auto var##_) { constexpr int var = decltype(var##_)::value; __VA_ARGS__ })

// value of v in lane L of the caller's 16-lane row (v_mov_b64_dpp row_newbcast:L)
template <int L>
__device__ __forceinline__ double bc(double v) {
    const long long x = __builtin_bit_cast(long long, v);
    const long long r = __builtin_amdgcn_update_dpp((long long)0, x, 0x150 + L, 0xf, 0xf, true);
    return __builtin_bit_cast(double, r);
}
__device__ __forceinline__ double row_sum(double x) {
    double s = 0.0;
    SFOR(l, 0, 16, { s += bc<l>(x); });
    return s;
}
__device__ __forceinline__ double row_min(double x) {
    double s = x;
    SFOR(l, 0, 16, { s = fmin(s, bc<l>(x)); });
    return s;
}
__device__ __forceinline__ double row_max(double x) {
    double s = x;
    SFOR(l, 0, 16, { s = fmax(s, bc<l>(x)); });
    return s;
}

struct Lane {
    int L;      // lane in row: 0..12 state rows, 13 affine row, 14/15 idle
    int q;      // instance in wave 0..3
    int wave;   // wave (= workgroup) index
    int inst;   // global instance
    bool valid;
};
__device__ __forceinline__ Lane lane_id(const Params& P) {
    Lane t;
    t.L = threadIdx.x & 15;
    t.q = threadIdx.x >> 4;
    t.wave = blockIdx.x;
    t.inst = t.wave * 4 + t.q;
    t.valid = t.inst < P.B;
    return t;
}
// Workspace pointers live inside the by-value Params struct, where clang cannot infer the
// address space: gm() re-types them as global (address_space(1)) so that loads / stores are
// global_* instead of flat_*.
typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) int gint;
__device__ __forceinline__ gdouble* gm(double* p) { return (gdouble*)(unsigned long long)p; }
__device__ __forceinline__ const gdouble* gm(const double* p) { return (const gdouble*)(unsigned long long)p; }
__device__ __forceinline__ gint* gm(int* p) { return (gint*)(unsigned long long)p; }
// (wave, stage) block of a field
__device__ __forceinline__ gdouble* blk(double* f, const Lane& t, int nst, int k, int sz) {
    return gm(f) + ((size_t)t.wave * nst + k) * sz;
}
// hides a value from common-subexpression elimination (keeps broadcast temporaries short-lived)
__device__ __forceinline__ void opaque(double& x) { asm volatile("" : "+v"(x)); }
// a value produced by one of the asm primitives of cfnmpc_dpp.hpp is about to be read through
// DPP by compiler-generated code (bc<>): give it the two wait states hipcc cannot know about
__device__ __forceinline__ void settle(double& x) { asm volatile("s_nop 1" : "+v"(x)); }
// instance-major 4-vectors (interior-point state, inputs): [inst][stage][4]
__device__ __forceinline__ size_t i4(const Params& P, const Lane& t, int k, int a) {
    return ((size_t)t.inst * P.N + k) * 4 + a;
}

// Loads are branch-free: every lane reads a valid (clamped) address and lanes outside the
// stored range select 0 -- exec-masked loads would split the unrolled code into tiny blocks.
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ double ld13(const gdouble* b, const Lane& t) {
    const double v = b[t.q * 13 + imin(t.L, 12)];
    return t.L < 13 ? v : 0.0;
}
__device__ __forceinline__ void st13(gdouble* b, const Lane& t, double v) { if (t.L < 13) b[t.q * 13 + t.L] = v; }

__device__ __forceinline__ void ld_ar(const gdouble* b, const Lane& t, double (&ar)[10]) {
    SFOR(s, 0, 10, {
        const double v = b[4 * ar_pre(s) + t.q * ar_n(s) + imin(t.L, ar_n(s) - 1)];
        ar[s] = t.L < ar_n(s) ? v : 0.0;
    });
}
__device__ __forceinline__ void ld_ac(const gdouble* b, const Lane& t, double (&ac)[13]) {
    SFOR(r, 0, 13, {
        const int c = imin(imax(t.L, ac_first(r)), 12);
        const double v = b[4 * ac_pre(r) + t.q * ac_m(r) + (c - ac_first(r))];
        ac[r] = (t.L >= ac_first(r) && t.L < 13) ? v : 0.0;
    });
}
__device__ __forceinline__ void ld_rows4(const gdouble* b, const Lane& t, double (&r)[4]) {  // BR / KP
    SFOR(a, 0, 4, {
        const double v = b[(a * 4 + t.q) * 13 + imin(t.L, 12)];
        r[a] = t.L < 13 ? v : 0.0;
    });
}
__device__ __forceinline__ void ld_cols4(const gdouble* b, const Lane& t, double (&c)[13]) {  // BC / KR
    SFOR(l, 0, 13, {
        const double v = b[(l * 4 + t.q) * 4 + (t.L & 3)];
        c[l] = t.L < 4 ? v : 0.0;
    });
}

// select element `idx` (runtime) of a register array
template <int N>
__device__ __forceinline__ double pick(const double (&a)[N], int idx) {
    double r = 0.0;
    SFOR(j, 0, N, { r = (idx == j) ? a[j] : r; });
    return r;
}

// =============================================================================================
// linearisation
// =============================================================================================
__global__ __launch_bounds__(64) void k_linearise(Params P) {
    __shared__ double tile[4][13 * 17 + 3];  // [instance][row r (internal)][17 columns: 13 x + 4 u]
    const Lane t = lane_id(P);
    const double h = P.dt;
    // this lane's sensitivity column: lanes 3..12 -> state column (internal index = lane),
    // lanes 0,1,2,13 -> input columns 0,1,2,3; lanes 14,15 idle.  (p columns are unit vectors.)
    const bool is_x = t.L >= 3 && t.L < 13;
    const bool is_u = t.L < 3 || t.L == 13;
    const int ucol = t.L == 13 ? 3 : t.L;
    const int xcol_ext = ext_of(is_x ? t.L : 3);
    const int tcol = is_x ? t.L : 13 + ucol;  // column in the LDS tile
    double* tl_ = tile[t.q];

    for (int k = 0; k < P.N; k++) {
        // every lane of the row holds the full (x_k, u_k) in EXTERNAL order (model code order)
        double x[13], u[4], xt[13], k1[13], k2[13], k3[13], k4[13];
        const gdouble* xb = blk(P.xit, t, P.N + 1, k, SZ_V13);
        SFOR(e, 0, 13, { x[e] = xb[t.q * 13 + int_of(e)]; });
        SFOR(a, 0, 4, { u[a] = gm(P.uit)[i4(P, t, k, a)]; });
        JacPoint J0, J1, J2, J3;
        f_expl(x, u, k1);
        jac_point(x, J0);
        SFOR(e, 0, 13, { xt[e] = x[e] + 0.5 * h * k1[e]; });
        f_expl(xt, u, k2);
        jac_point(xt, J1);
        SFOR(e, 0, 13, { xt[e] = x[e] + 0.5 * h * k2[e]; });
        f_expl(xt, u, k3);
        jac_point(xt, J2);
        SFOR(e, 0, 13, { xt[e] = x[e] + h * k3[e]; });
        f_expl(xt, u, k4);
        jac_point(xt, J3);
        double phi[13];
        SFOR(e, 0, 13, { phi[e] = x[e] + (h / 6.0) * (k1[e] + 2 * k2[e] + 2 * k3[e] + k4[e]); });
        // b = Phi - x_{k+1}, distributed (lane i <-> internal state i)
        {
            const double xn = ld13(blk(P.xit, t, P.N + 1, k + 1, SZ_V13), t);
            const double ph = pick(phi, ext_of(t.L < 13 ? t.L : 0));
            st13(blk(P.b, t, P.N, k, SZ_V13), t, ph - xn);
        }
        // this lane's sensitivity column through the four RK stages
        double s0[13], s[13], c1[13], c2[13], c3[13], c4[13], ju[4] = {0, 0, 0, 0};
        SFOR(e, 0, 13, { s0[e] = (is_x && xcol_ext == e) ? 1.0 : 0.0; });
        if (is_u) {
            const double uc = 2.0 * pick(u, ucol);
            const double sa = (ucol < 2) ? 1.0 : -1.0;
            const double sb = (ucol == 0 || ucol == 3) ? 1.0 : -1.0;
            const double sc = (ucol == 0 || ucol == 2) ? 1.0 : -1.0;
            ju[0] = KT * uc; ju[1] = KA * sa * uc; ju[2] = KB * sb * uc; ju[3] = KC * sc * uc;
        }
        jvp<true, true>(J0, s0, c1);
        SFOR(i, 0, 4, { c1[9 + i] += ju[i]; });
        SFOR(e, 0, 13, { s[e] = s0[e] + 0.5 * h * c1[e]; });
        jvp<true, true>(J1, s, c2);
        SFOR(i, 0, 4, { c2[9 + i] += ju[i]; });
        SFOR(e, 0, 13, { s[e] = s0[e] + 0.5 * h * c2[e]; });
        jvp<true, true>(J2, s, c3);
        SFOR(i, 0, 4, { c3[9 + i] += ju[i]; });
        SFOR(e, 0, 13, { s[e] = s0[e] + h * c3[e]; });
        jvp<true, true>(J3, s, c4);
        SFOR(i, 0, 4, { c4[9 + i] += ju[i]; });
        double col[13];  // internal row order
        SFOR(r, 0, 13, {
            constexpr int e = ext_of(r);
            col[r] = s0[e] + (h / 6.0) * (c1[e] + 2 * c2[e] + 2 * c3[e] + c4[e]);
        });
        // column form of A straight from registers
        {
            gdouble* ac = blk(P.AC, t, P.N, k, SZ_A);
            SFOR(r, 0, 13, {
                if (is_x && t.L >= ac_first(r)) ac[4 * ac_pre(r) + t.q * ac_m(r) + (t.L - ac_first(r))] = col[r];
            });
            gdouble* bcb = blk(P.BC, t, P.N, k, SZ_B);
            SFOR(r, 0, 13, { if (is_u) bcb[(r * 4 + t.q) * 4 + ucol] = col[r]; });
        }
        // row forms through the LDS tile
        __syncthreads();
        if (is_x || is_u) SFOR(r, 0, 13, { tl_[r * 17 + tcol] = col[r]; });
        __syncthreads();
        {
            gdouble* ar = blk(P.AR, t, P.N, k, SZ_A);
            SFOR(sl, 0, 10, { if (t.L < ar_n(sl)) ar[4 * ar_pre(sl) + t.q * ar_n(sl) + t.L] = tl_[t.L * 17 + (sl + 3)]; });
            gdouble* brb = blk(P.BR, t, P.N, k, SZ_B);
            SFOR(a, 0, 4, { if (t.L < 13) brb[(a * 4 + t.q) * 13 + t.L] = tl_[t.L * 17 + 13 + a]; });
        }
    }
}

// =============================================================================================
// Riccati sweeps
// =============================================================================================
// symmetric positive definite 4x4 (packed upper) -> inverse (packed upper); false if not SPD
__device__ __forceinline__ bool spd4_inv(const double (&S)[10], double (&Si)[10]) {
    double Lm[4][4], Li[4][4];
    bool ok = true;
    SFOR(j, 0, 4, {
        double s = S[s4(j, j)];
        SFOR(k, 0, j, { s -= Lm[j][k] * Lm[j][k]; });
        ok = ok && (s > 0.0);
        const double ljj = sqrt(s);
        const double inv = 1.0 / ljj;
        Lm[j][j] = ljj;
        Li[j][j] = inv;
        SFOR(i, j + 1, 4, {
            double tt = S[s4(i, j)];
            SFOR(k, 0, j, { tt -= Lm[i][k] * Lm[j][k]; });
            Lm[i][j] = tt * inv;
        });
    });
    SFOR(j, 0, 4, {
        SFOR(i, j + 1, 4, {
            double tt = 0;
            SFOR(k, j, i, { tt -= Lm[i][k] * Li[k][j]; });
            Li[i][j] = tt * Li[i][i];
        });
    });
    SFOR(i, 0, 4, {
        SFOR(j, i, 4, {
            double tt = 0;
            SFOR(k, j, 4, { tt += Li[k][i] * Li[k][j]; });
            Si[s4(i, j)] = tt;
        });
    });
    return ok;
}

// One stage of the augmented backward recursion.
//   Pa[13]: lanes 0..12 row i of P_{k+1}; lane 13 the affine row p_{k+1}' (delta form) -- on
//           exit the same for stage k.
//   ABSOLUTE: start solve with the QP's own affine terms (q_k, b_k, r_k);
//   otherwise input Hessian R^ and gradient g are read from P.Rh / P.g (interior-point step).
template <bool ABSOLUTE>
__device__ __forceinline__ bool factor_stage(const Params& P, const Lane& t, const int k, double (&Pa)[13],
                                             double* wt /* LDS [13*17] of this instance */) {
    double ar[10], br[4], bcl[13];
    ld_ar(blk(P.AR, t, P.N, k, SZ_A), t, ar);
    ld_rows4(blk(P.BR, t, P.N, k, SZ_B), t, br);
    ld_cols4(blk(P.BC, t, P.N, k, SZ_B), t, bcl);
    // input Hessian / gradient: lane a < 4 holds element a
    double Rh, g;
    {
        const int a = t.L & 3;
        if (ABSOLUTE) {
            const double uk = gm(P.uit)[i4(P, t, k, a)];
            const double yr = blk(P.yref, t, P.N, k, SZ_Y)[t.q * 17 + 13 + a];
            Rh = P.W[13 + a];
            g = P.W[13 + a] * (uk - yr);
        } else {
            Rh = gm(P.Rh)[i4(P, t, k, a)];
            g = gm(P.g)[i4(P, t, k, a)];
        }
        if (t.L >= 4) { Rh = 0.0; g = 0.0; }
    }
    if (ABSOLUTE) {
        // hb' = p' + (P b)' in lane 13
        const double bv = ld13(blk(P.b, t, P.N, k, SZ_V13), t);
        double pb = 0.0;
        dotbc<13, 0>(pb, Pa, bv);
        if (t.L >= 13) pb = 0.0;
        SFOR(j, 0, 13, {
            const double add = bc<j>(pb);
            if (t.L == 13) Pa[j] += add;
        });
    }
    // (1) W = Pa A (row form, instruction-level sparsity of A), (2) V = Pa B
    double W[13], V[4];
    SFOR(j, 0, 3, { W[j] = Pa[j]; });
    SFOR(j, 3, 13, { W[j] = 0.0; });
    dot2bc<6, 0>(W[3], W[4], Pa, ar[0], ar[1]);
    dotbc<6, 0>(W[5], Pa, ar[2]);
    dot2bc<10, 0>(W[6], W[7], Pa, ar[3], ar[4]);
    dot2bc<10, 0>(W[8], W[9], Pa, ar[5], ar[6]);
    dot2bc<13, 0>(W[10], W[11], Pa, ar[7], ar[8]);
    dotbc<13, 0>(W[12], Pa, ar[9]);
    SFOR(a, 0, 4, { V[a] = 0.0; });
    dot2bc<13, 0>(V[0], V[1], Pa, br[0], br[1]);
    dot2bc<13, 0>(V[2], V[3], Pa, br[2], br[3]);
    // (3) Wt = transpose of W over lanes 0..12; lane 13 keeps the affine row
    double Wt[13];
    __syncthreads();
    if (t.L < 13) SFOR(j, 0, 13, { wt[j * 17 + t.L] = W[j]; });
    __syncthreads();
    SFOR(l, 0, 13, {
        const double w = wt[imin(t.L, 12) * 17 + l];
        Wt[l] = t.L < 13 ? w : (t.L == 13 ? Pa[l] : 0.0);
    });
    // (4) M = Q + Wt A  (lane 13: q_k' + hb'A)
    double M[13];
    if (ABSOLUTE) {
        // q_k distributed -> row form in lane 13
        const double xk = ld13(blk(P.xit, t, P.N + 1, k, SZ_V13), t);
        const double yk = blk(P.yref, t, P.N, k, SZ_Y)[t.q * 17 + imin(t.L, 12)];
        double qv = 0.0;
        SFOR(j, 0, 13, { if (t.L == j) qv = P.W[ext_of(j)] * (xk - yk); });
        SFOR(j, 0, 13, {
            const double qj = bc<j>(qv);
            M[j] = (t.L == j) ? P.W[ext_of(j)] : (t.L == 13 ? qj : 0.0);
        });
    } else {
        SFOR(j, 0, 13, { M[j] = (t.L == j) ? P.W[ext_of(j)] : 0.0; });
    }
    SFOR(j, 0, 3, { M[j] += Wt[j]; });
    dot2bc<6, 0>(M[3], M[4], Wt, ar[0], ar[1]);
    dotbc<6, 0>(M[5], Wt, ar[2]);
    dot2bc<10, 0>(M[6], M[7], Wt, ar[3], ar[4]);
    dot2bc<10, 0>(M[8], M[9], Wt, ar[5], ar[6]);
    dot2bc<13, 0>(M[10], M[11], Wt, ar[7], ar[8]);
    dotbc<13, 0>(M[12], Wt, ar[9]);
    // (5) G' = Wt B ; lane 13: rho = g + B'hb
    double Gp[4];
    SFOR(a, 0, 4, { Gp[a] = 0.0; });
    dot2bc<13, 0>(Gp[0], Gp[1], Wt, br[0], br[1]);
    dot2bc<13, 0>(Gp[2], Gp[3], Wt, br[2], br[3]);
    SFOR(a, 0, 4, {
        const double ga = bc<a>(g);
        if (t.L == 13) Gp[a] += ga;
    });
    // (6) S = R^ + B'V in lanes a < 4, replicated, inverted redundantly by every lane
    double Srow[4];
    SFOR(c, 0, 4, { Srow[c] = (t.L == c) ? Rh : 0.0; });
    dot2bc<13, 0>(Srow[0], Srow[1], bcl, V[0], V[1]);
    dot2bc<13, 0>(Srow[2], Srow[3], bcl, V[2], V[3]);
    SFOR(c, 0, 4, { settle(Srow[c]); });
    double S[10], Si[10];
    SFOR(a, 0, 4, { SFOR(c, a, 4, { S[s4(a, c)] = bc<a>(Srow[c]); }); });
    const bool ok = spd4_inv(S, Si);
    // (7) K' = G' Sinv  (lane 13: feed-forward d)
    double Kp[4], nGp[4];
    SFOR(a, 0, 4, {
        double s = 0.0;
        SFOR(c, 0, 4, { s += Gp[c] * Si[s4(c, a)]; });
        Kp[a] = s;
        nGp[a] = -Gp[a];
    });
    // (8) P <- M - G' K  (lane 13: p' <- M_13 - rho' K)
    SFOR(j, 0, 13, {
        Pa[j] = M[j];
        updbc<j>(Pa[j], Kp, nGp);
    });
    // (9) stores
    {
        gdouble* kp = blk(P.KP, t, P.N, k, SZ_K);
        gdouble* kr = blk(P.KR, t, P.N, k, SZ_K);
        SFOR(a, 0, 4, {
            if (t.L < 13) {
                kp[(a * 4 + t.q) * 13 + t.L] = Kp[a];
                kr[(t.L * 4 + t.q) * 4 + a] = Kp[a];
            }
            if (t.L == 13) gm(P.d)[i4(P, t, k, a)] = Kp[a];
        });
        if (t.L == 0) {
            gdouble* sv = blk(P.Sinv, t, P.N, k, SZ_S);
            SFOR(e, 0, 10, { sv[t.q * 10 + e] = Si[e]; });
        }
    }
    return ok;
}

// Backward factorisation over stages [0, head).  If FROM_CHK the recursion starts from a stored
// checkpoint of the unconstrained tail (P.Pchk, affine row zero), else from the terminal cost.
template <bool ABSOLUTE>
__device__ __forceinline__ bool sweep_factor(const Params& P, const Lane& t, const int head, const int chk,
                                             double* wt) {
    double Pa[13];
    if (ABSOLUTE || chk < 0) {
        const double xN = ABSOLUTE ? ld13(blk(P.xit, t, P.N + 1, P.N, SZ_V13), t) : 0.0;
        const double yN = ABSOLUTE ? ld13(blk(P.yref_e, t, 1, 0, SZ_V13), t) : 0.0;
        double qv = 0.0;
        SFOR(j, 0, 13, { if (t.L == j) qv = P.WN[ext_of(j)] * (xN - yN); });
        SFOR(j, 0, 13, {
            const double qj = bc<j>(qv);
            Pa[j] = (t.L == j) ? P.WN[ext_of(j)] : ((ABSOLUTE && t.L == 13) ? qj : 0.0);
        });
    } else {
        const gdouble* pc = gm(P.Pchk) + ((size_t)t.wave * N_CHK + chk) * SZ_P;
        SFOR(j, 0, 13, {
            const double v = pc[(j * 4 + t.q) * 13 + imin(t.L, 12)];
            Pa[j] = t.L < 13 ? v : 0.0;
        });
    }
    bool ok = true;
    for (int k = head - 1; k >= 0; k--) {
        ok = factor_stage<ABSOLUTE>(P, t, k, Pa, wt) && ok;
        if (ABSOLUTE) {
            // checkpoints of the unconstrained cost-to-go (matrix part only)
            SFOR(c, 0, N_CHK, {
                if (k == chk_stage(c)) {
                    gdouble* pc = gm(P.Pchk) + ((size_t)t.wave * N_CHK + c) * SZ_P;
                    SFOR(j, 0, 13, { if (t.L < 13) pc[(j * 4 + t.q) * 13 + t.L] = Pa[j]; });
                }
            });
        }
    }
    return ok;
}

// x+ = A x + B v (+ b), all distributed; vr[4] replicated
template <bool WITH_B>
__device__ __forceinline__ double propagate(const Params& P, const Lane& t, const int k, const double x,
                                            const double (&vr)[4]) {
    double ar[10], br[4];
    ld_ar(blk(P.AR, t, P.N, k, SZ_A), t, ar);
    ld_rows4(blk(P.BR, t, P.N, k, SZ_B), t, br);
    double xn = t.L < 3 ? x : 0.0;
    if (WITH_B) xn += ld13(blk(P.b, t, P.N, k, SZ_V13), t);
    dotbc<10, 3>(xn, ar, x);
    SFOR(a, 0, 4, { xn += br[a] * vr[a]; });
    return xn;
}

// v = -K x - d in lanes a < 4
__device__ __forceinline__ double feedback(const Params& P, const Lane& t, const int k, const double x) {
    double kr[13];
    ld_cols4(blk(P.KR, t, P.N, k, SZ_K), t, kr);
    const double dk = gm(P.d)[i4(P, t, k, t.L & 3)];
    double v = t.L < 4 ? -dk : 0.0;
    double acc = 0.0;
    dotbc<13, 0>(acc, kr, x);
    v -= acc;
    settle(v);
    return v;
}

// forward sweep of a homogeneous (delta) solve over [0, head): writes the input step to `out`
__device__ __forceinline__ void sweep_forward_delta(const Params& P, const Lane& t, const int head, gdouble* out) {
    double x = 0.0;
    for (int k = 0; k < head; k++) {
        const double dv = feedback(P, t, k, x);
        if (t.L < 4) out[i4(P, t, k, t.L)] = dv;
        double vr[4];
        SFOR(a, 0, 4, { vr[a] = bc<a>(dv); });
        if (k + 1 < head) x = propagate<false>(P, t, k, x, vr);
    }
}

// backward sweep re-using the factorisation for the right-hand side P.g (input rows only);
// overwrites d
__device__ __forceinline__ void sweep_resolve(const Params& P, const Lane& t, const int head) {
    double p = 0.0;
    for (int k = head - 1; k >= 0; k--) {
        double bcl[13], ac[13], kp[4];
        ld_cols4(blk(P.BC, t, P.N, k, SZ_B), t, bcl);
        ld_ac(blk(P.AC, t, P.N, k, SZ_A), t, ac);
        ld_rows4(blk(P.KP, t, P.N, k, SZ_K), t, kp);
        const int a = t.L & 3;
        const double gk = gm(P.g)[i4(P, t, k, a)];
        double rho = t.L < 4 ? gk : 0.0;
        dotbc<13, 0>(rho, bcl, p);
        settle(rho);
        double rr[4];
        SFOR(c, 0, 4, { rr[c] = bc<c>(rho); });
        const gdouble* sv = blk(P.Sinv, t, P.N, k, SZ_S) + t.q * 10;
        double dd = 0.0;
        SFOR(c, 0, 4, {
            const int lo = a < c ? a : c, hi = a < c ? c : a;
            dd += sv[lo * 4 - (lo * (lo - 1)) / 2 + (hi - lo)] * rr[c];
        });
        if (t.L < 4) gm(P.d)[i4(P, t, k, a)] = dd;
        double pn = t.L < 3 ? p : 0.0;
        dotbc<13, 0>(pn, ac, p);
        SFOR(c, 0, 4, { pn -= kp[c] * rr[c]; });
        p = pn;
    }
}

// =============================================================================================
// QP solve + RTI update
// =============================================================================================
struct RowIPM {  // uniform over the 16 lanes of a row
    double mu, res;
    int iters, status;
    bool act;
};

__device__ __forceinline__ double ratio(double z, double dz, double a) {
    const double tt = -z / dz;
    return (dz < 0.0 && tt < a) ? tt : a;
}

// element-wise passes: lane L of a row handles elements e = L, L+16, ... of the head*4 inputs
struct Elem {
    double v, tl, tu, ll, lu, rg, lb, ub;
};
__device__ __forceinline__ Elem ld_elem(const Params& P, const Lane& t, size_t idx) {
    Elem e;
    e.v = gm(P.v)[idx]; e.tl = gm(P.tl)[idx]; e.tu = gm(P.tu)[idx]; e.ll = gm(P.ll)[idx]; e.lu = gm(P.lu)[idx]; e.rg = gm(P.rg)[idx];
    const double uk = gm(P.uit)[idx];
    e.lb = P.u_min - uk;
    e.ub = P.u_max - uk;
    return e;
}

__global__ __launch_bounds__(64) void k_qp(Params P) {
    __shared__ double wtile[4][13 * 17 + 3];
    const Lane t = lane_id(P);
    double* wt = wtile[t.q];
    const int N = P.N;
    const size_t ibase = (size_t)t.inst * N * 4;  // this instance's 4-vectors
    RowIPM R;
    R.iters = 0; R.status = 0; R.res = 0.0; R.mu = 0.0; R.act = false;

    // ---- start: unconstrained minimiser (absolute Riccati solve over the whole horizon)
    bool ok = sweep_factor<true>(P, t, N, -1, wt);
    double viol = 0.0;      // max bound violation of the unconstrained inputs
    int last_tight = -1;    // last stage whose unconstrained input is outside / near a bound
    {
        const double margin = 0.05 * (P.u_max - P.u_min);
        bool sawnan = false;
        double x = ld13(blk(P.x0, t, 1, 0, SZ_V13), t) - ld13(blk(P.xit, t, N + 1, 0, SZ_V13), t);
        for (int k = 0; k < N; k++) {
            const double v = feedback(P, t, k, x);
            if (t.L < 4) {
                const double uk = gm(P.uit)[i4(P, t, k, t.L)];
                const double lb = P.u_min - uk, ub = P.u_max - uk;
                gm(P.v)[i4(P, t, k, t.L)] = v;
                viol = fmax(viol, fmax(lb - v, v - ub));
                if (v < lb + margin || v > ub - margin) last_tight = k;
                sawnan = sawnan || !(v == v);
            }
            double vr[4];
            SFOR(a, 0, 4, { vr[a] = bc<a>(v); });
            x = propagate<true>(P, t, k, x, vr);
        }
        viol = row_max(viol);
        last_tight = (int)row_max((double)last_tight);
        if (row_max(sawnan ? 1.0 : 0.0) > 0.0) viol = nan("");
    }
    ok = row_min(ok ? 1.0 : 0.0) > 0.0;

    // ---- head of the horizon the interior-point sweeps work on (wave-uniform)
    int head = N, chk = -1;
    const bool infeasible = t.valid && ok && (viol > 0.0);
    if (P.active_horizon) {
        int want = infeasible ? last_tight + 3 : 0;
        // wave-uniform maximum over the four instances
        want = max(want, __shfl_xor(want, 16));
        want = max(want, __shfl_xor(want, 32));
        head = N;
        SFOR(c, 0, N_CHK, {
            constexpr int cs = chk_stage(N_CHK - 1 - c);
            if (want <= cs && cs < N) { head = cs; chk = N_CHK - 1 - c; }
        });
    }

    for (int attempt = 0; attempt < 2; attempt++) {
        if (!t.valid) {
            R.status = 0;
        } else if (!ok || !(viol == viol)) {
            R.status = 4;
            R.res = nan("");
        } else if (infeasible) {
            // ---- shift slacks / multipliers positive; residuals of the start; first R^, g
            const double mu0 = fmax(viol, P.lam0_min);
            double mu = 0.0, res = 0.0;
            for (int e = t.L; e < head * 4; e += 16) {
                const size_t idx = ibase + e;
                const double uk = gm(P.uit)[idx], v = gm(P.v)[idx];
                const double lb = P.u_min - uk, ub = P.u_max - uk;
                const double tl = fmax(v - lb, P.thr0), tu = fmax(ub - v, P.thr0);
                const double ll = mu0 / tl, lu = mu0 / tu, rg = -ll + lu;
                gm(P.tl)[idx] = tl; gm(P.tu)[idx] = tu; gm(P.ll)[idx] = ll; gm(P.lu)[idx] = lu; gm(P.rg)[idx] = rg;
                const double rl = v - lb - tl, ru = ub - v - tu;
                const double Dl = ll / tl, Du = lu / tu;
                gm(P.Rh)[idx] = P.W[13 + (e & 3)] + Dl + Du;
                gm(P.g)[idx] = rg + ll + Dl * rl - lu - Du * ru;
                mu += ll * tl + lu * tu;
                res = fmax(res, fmax(fmax(ll * tl, lu * tu), fmax(fabs(rg), fmax(fabs(rl), fabs(ru)))));
            }
            R.mu = row_sum(mu) / (8.0 * head);
            R.res = row_max(res);
            R.act = true;
            R.status = 2;
        }

        // ---- interior-point loop, wave-uniform trip count
        while (__any(R.act)) {
            if (R.act) {
                if (!(R.res == R.res)) { R.status = 4; R.act = false; }
                else if (R.res <= P.tol) { R.status = 0; R.act = false; }
                else if (R.iters >= P.max_iter) { R.status = 2; R.act = false; }
            }
            if (!__any(R.act)) break;
            if (R.act) R.iters++;
            // predictor: factorise (R^, g from the element-wise pass), forward
            const bool fok = sweep_factor<false>(P, t, head, chk, wt);
            sweep_forward_delta(P, t, head, gm(P.dva));
            // affine step length, mu_aff, centering; corrector right-hand side
            double smu;
            {
                double a = 1.0;
                for (int e = t.L; e < head * 4; e += 16) {
                    const Elem el = ld_elem(P, t, ibase + e);
                    const double dva = gm(P.dva)[ibase + e];
                    const double rl = el.v - el.lb - el.tl, ru = el.ub - el.v - el.tu;
                    const double dtl = dva + rl, dtu = -dva + ru;
                    const double dll = -el.ll - (el.ll / el.tl) * dtl, dlu = -el.lu - (el.lu / el.tu) * dtu;
                    a = ratio(el.tl, dtl, a); a = ratio(el.tu, dtu, a);
                    a = ratio(el.ll, dll, a); a = ratio(el.lu, dlu, a);
                }
                a = row_min(a);
                double mu_aff = 0.0;
                for (int e = t.L; e < head * 4; e += 16) {
                    const Elem el = ld_elem(P, t, ibase + e);
                    const double dva = gm(P.dva)[ibase + e];
                    const double rl = el.v - el.lb - el.tl, ru = el.ub - el.v - el.tu;
                    const double dtl = dva + rl, dtu = -dva + ru;
                    const double dll = -el.ll - (el.ll / el.tl) * dtl, dlu = -el.lu - (el.lu / el.tu) * dtu;
                    mu_aff += (el.ll + a * dll) * (el.tl + a * dtl) + (el.lu + a * dlu) * (el.tu + a * dtu);
                }
                mu_aff = row_sum(mu_aff) / (8.0 * head);
                const double sr = mu_aff / R.mu;
                smu = sr * sr * sr * R.mu;
                for (int e = t.L; e < head * 4; e += 16) {
                    const Elem el = ld_elem(P, t, ibase + e);
                    const double dva = gm(P.dva)[ibase + e];
                    const double rl = el.v - el.lb - el.tl, ru = el.ub - el.v - el.tu;
                    const double dtl = dva + rl, dtu = -dva + ru;
                    const double dll = -el.ll - (el.ll / el.tl) * dtl, dlu = -el.lu - (el.lu / el.tu) * dtu;
                    const double cl = dll * dtl, cu = dlu * dtu;
                    gm(P.g)[ibase + e] = (cl - smu) / el.tl - (cu - smu) / el.tu;
                }
            }
            // corrector: re-solve, forward
            sweep_resolve(P, t, head);
            sweep_forward_delta(P, t, head, gm(P.dvc));
            // step, update, residuals of the new point, next R^ and g
            {
                double a = 1.0;
                for (int e = t.L; e < head * 4; e += 16) {
                    const Elem el = ld_elem(P, t, ibase + e);
                    const double dva = gm(P.dva)[ibase + e], dv = dva + gm(P.dvc)[ibase + e];
                    const double rl = el.v - el.lb - el.tl, ru = el.ub - el.v - el.tu;
                    const double dtla = dva + rl, dtua = -dva + ru;
                    const double Dl = el.ll / el.tl, Du = el.lu / el.tu;
                    const double cl = (-el.ll - Dl * dtla) * dtla, cu = (-el.lu - Du * dtua) * dtua;
                    const double dtl = dv + rl, dtu = -dv + ru;
                    const double dll = (smu - cl) / el.tl - el.ll - Dl * dtl, dlu = (smu - cu) / el.tu - el.lu - Du * dtu;
                    a = ratio(el.tl, dtl, a); a = ratio(el.tu, dtu, a);
                    a = ratio(el.ll, dll, a); a = ratio(el.lu, dlu, a);
                }
                a = fmin(1.0, P.tau * row_min(a));
                double mu = 0.0, res = 0.0;
                for (int e = t.L; e < head * 4; e += 16) {
                    const size_t idx = ibase + e;
                    const Elem el = ld_elem(P, t, idx);
                    const double dva = gm(P.dva)[idx], dv = dva + gm(P.dvc)[idx];
                    const double rl = el.v - el.lb - el.tl, ru = el.ub - el.v - el.tu;
                    const double dtla = dva + rl, dtua = -dva + ru;
                    const double Dl0 = el.ll / el.tl, Du0 = el.lu / el.tu;
                    const double cl = (-el.ll - Dl0 * dtla) * dtla, cu = (-el.lu - Du0 * dtua) * dtua;
                    const double dtl = dv + rl, dtu = -dv + ru;
                    const double dll = (smu - cl) / el.tl - el.ll - Dl0 * dtl, dlu = (smu - cu) / el.tu - el.lu - Du0 * dtu;
                    const double v = el.v + a * dv, tl = el.tl + a * dtl, tu = el.tu + a * dtu;
                    const double ll = el.ll + a * dll, lu = el.lu + a * dlu, rg = el.rg * (1.0 - a);
                    const double rln = v - el.lb - tl, run = el.ub - v - tu;
                    const double Dl = ll / tl, Du = lu / tu;
                    if (R.act) {
                        gm(P.v)[idx] = v; gm(P.tl)[idx] = tl; gm(P.tu)[idx] = tu; gm(P.ll)[idx] = ll; gm(P.lu)[idx] = lu; gm(P.rg)[idx] = rg;
                        gm(P.Rh)[idx] = P.W[13 + (e & 3)] + Dl + Du;
                        gm(P.g)[idx] = rg + ll + Dl * rln - lu - Du * run;
                    }
                    mu += ll * tl + lu * tu;
                    res = fmax(res, fmax(fmax(ll * tl, lu * tu), fmax(fabs(rg), fmax(fabs(rln), fabs(run)))));
                }
                mu = row_sum(mu) / (8.0 * head);
                res = row_max(res);
                const bool fok_row = row_min(fok ? 1.0 : 0.0) > 0.0;
                if (R.act) {
                    R.mu = mu;
                    R.res = fok_row ? res : nan("");
                }
            }
        }

        // ---- expand: dynamics-exact roll-out; head stages use the QP inputs, tail stages the
        //      unconstrained feedback law of the start solve, whose inputs must stay inside the box
        bool tail_ok = true;
        {
            double x = ld13(blk(P.x0, t, 1, 0, SZ_V13), t) - ld13(blk(P.xit, t, N + 1, 0, SZ_V13), t);
            for (int k = 0; k < N; k++) {
                st13(blk(P.dx, t, N + 1, k, SZ_V13), t, x);
                double v;
                if (k < head) {
                    const double vk = gm(P.v)[i4(P, t, k, t.L & 3)];
                    v = t.L < 4 ? vk : 0.0;
                } else {
                    v = feedback(P, t, k, x);
                    if (t.L < 4) {
                        const double uk = gm(P.uit)[i4(P, t, k, t.L)];
                        tail_ok = tail_ok && (v >= P.u_min - uk) && (v <= P.u_max - uk);
                        gm(P.v)[i4(P, t, k, t.L)] = v;
                    }
                }
                double vr[4];
                SFOR(a, 0, 4, { vr[a] = bc<a>(v); });
                x = propagate<true>(P, t, k, x, vr);
            }
            st13(blk(P.dx, t, N + 1, N, SZ_V13), t, x);
            tail_ok = row_min(tail_ok ? 1.0 : 0.0) > 0.0;
        }
        const bool redo = t.valid && R.status != 4 && !tail_ok && head < N;
        if (!__any(redo)) break;
        // rare: a tail input left the box -> solve again over the full horizon (whole wave)
        head = N; chk = -1;
        R.iters = 0; R.status = 0; R.res = 0.0; R.act = false;
        ok = sweep_factor<true>(P, t, N, -1, wt);
        ok = row_min(ok ? 1.0 : 0.0) > 0.0;
        {
            double x = ld13(blk(P.x0, t, 1, 0, SZ_V13), t) - ld13(blk(P.xit, t, N + 1, 0, SZ_V13), t);
            for (int k = 0; k < N; k++) {
                const double v = feedback(P, t, k, x);
                if (t.L < 4) gm(P.v)[i4(P, t, k, t.L)] = v;
                double vr[4];
                SFOR(a, 0, 4, { vr[a] = bc<a>(v); });
                x = propagate<true>(P, t, k, x, vr);
            }
        }
    }

    // ---- full RTI step (iterate += step) and statistics
    if (t.valid) {
        if (R.status != 4) {
            for (int k = 0; k <= N; k++) {
                gdouble* xb = blk(P.xit, t, N + 1, k, SZ_V13);
                const double dxk = ld13(blk(P.dx, t, N + 1, k, SZ_V13), t);
                if (t.L < 13) xb[t.q * 13 + t.L] += dxk;
            }
            for (int e = t.L; e < N * 4; e += 16) gm(P.uit)[ibase + e] += gm(P.v)[ibase + e];
        }
        if (t.L == 0) {
            gm(P.status)[t.inst] = R.status;
            gm(P.iters)[t.inst] = R.iters;
            gm(P.res)[t.inst] = R.res;
            gm(P.head)[t.inst] = head;
        }
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

// AoS [B][S][E] (external order) -> wave-blocked [wave][S][inst 0..3][E]; if perm13 the first 13
// entries of a row are permuted to the internal state order.  E == 4 fields are instance-major
// ([inst][S][4]) and handled by the same formula with a different block shape.
__device__ __forceinline__ size_t blk_index(int i, int s, int e, int S, int E) {
    if (E == 4) return ((size_t)i * S + s) * 4 + e;
    return (((size_t)(i >> 2) * S + s) * 4 + (i & 3)) * E + e;
}
__global__ void k_put(int B, int S, int E, int perm13, const double* __restrict__ aos, double* __restrict__ blkp) {
    const size_t n = (size_t)B * S * E;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(idx % E);
        const int s = (int)((idx / E) % S);
        const int i = (int)(idx / ((size_t)E * S));
        const int ei = (perm13 && e < 13) ? int_of(e) : e;
        blkp[blk_index(i, s, ei, S, E)] = aos[idx];
    }
}
__global__ void k_get(int B, int S, int E, int perm13, int s0, int Stot, const double* __restrict__ blkp,
                      double* __restrict__ aos) {
    const size_t n = (size_t)B * S * E;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(idx % E);
        const int s = (int)((idx / E) % S);
        const int i = (int)(idx / ((size_t)E * S));
        const int ei = (perm13 && e < 13) ? int_of(e) : e;
        aos[idx] = blkp[blk_index(i, s0 + s, ei, Stot, E)];
    }
}
__global__ void k_init_iterate(Params P, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.B) return;
    // generate_c_code.py:58,135 / SURVEY App. D-3
    const double hov = sqrt((MQ * G0) / (4 * CT));
    for (int k = 0; k <= P.N; k++)
        for (int e = 0; e < 13; e++) {
            const double x0e = P.x0[blk_index(i, 0, int_of(e), 1, 13)];
            P.xit[blk_index(i, k, int_of(e), P.N + 1, 13)] = (mode == 1) ? x0e : (e == 3 ? 1.0 : 0.0);
        }
    for (int k = 0; k < P.N; k++)
        for (int e = 0; e < 4; e++) P.uit[blk_index(i, k, e, P.N, 4)] = (mode == 1) ? hov : 0.0;
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
void launch_linearise(const Params& P, hipStream_t st) {
    hipLaunchKernelGGL(k_linearise, dim3(P.NW), dim3(64), 0, st, P);
}
void launch_qp(const Params& P, hipStream_t st) { hipLaunchKernelGGL(k_qp, dim3(P.NW), dim3(64), 0, st, P); }
void launch_sim(int B, const double* x, const double* u, double T, int steps, double* xn, hipStream_t st) {
    hipLaunchKernelGGL(k_sim, dim3((B + 255) / 256), dim3(256), 0, st, B, x, u, T, steps, xn);
}
static inline int grid_for(size_t n) {
    size_t g = (n + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}
void launch_put(int B, int S, int E, int perm13, const double* aos, double* blkp, hipStream_t st) {
    hipLaunchKernelGGL(k_put, dim3(grid_for((size_t)B * S * E)), dim3(256), 0, st, B, S, E, perm13, aos, blkp);
}
void launch_get(int B, int S, int E, int perm13, int s0, int Stot, const double* blkp, double* aos, hipStream_t st) {
    hipLaunchKernelGGL(k_get, dim3(grid_for((size_t)B * S * E)), dim3(256), 0, st, B, S, E, perm13, s0, Stot, blkp, aos);
}
void launch_init_iterate(const Params& P, int mode, hipStream_t st) {
    hipLaunchKernelGGL(k_init_iterate, dim3((P.B + 255) / 256), dim3(256), 0, st, P, mode);
}

}  // namespace cfn
