// cfnmpc_rg.hpp -- device primitives of the row-group mapping shared by the kernel translation units
// (cfnmpc_kernels.hip, cfnmpc_linfactor.hip): lane identity, DPP broadcasts / row reductions, wave-blocked
// block addressing and loads, the 4 x 4 Cholesky and ONE STAGE of the augmented backward Riccati recursion
// (factor_stage).  Layout and mapping: cfnmpc_ws.hpp; the broadcast-FMA primitives: cfnmpc_dpp.hpp.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "cfnmpc_dpp.hpp"
#include "cfnmpc_model.hpp"
#include "cfnmpc_ws.hpp"

#ifndef KALIGN_BYTES
#define KALIGN_BYTES 4096
#endif
#define KALIGN __attribute__((aligned(KALIGN_BYTES)))
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
// lane i of every row <- lane i + 4 (row_shl:4; lanes 12..15 get 0)
__device__ __forceinline__ double shift4(const double v) {
    const long long x = __builtin_bit_cast(long long, v);
    const long long r = __builtin_amdgcn_update_dpp((long long)0, x, 0x104, 0xf, 0xf, true);
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

__device__ __forceinline__ double lane_wu(const Params& P, int a) {
    double w = a == 0 ? P.W[13] : (a == 1 ? P.W[14] : (a == 2 ? P.W[15] : P.W[16]));
    asm volatile("" : "+v"(w));
    return w;
}
struct Lane {
    int L;      // lane in row: 0..12 state rows, 13 affine row, 14/15 idle
    int row;    // DPP row of this lane inside the wavefront, 0..3 (LDS tile index)
    int q;      // position of the instance inside its workspace block, 0..3
    int wave;   // workspace block (= "home" wave) of the instance
    int inst;   // global instance
    bool valid;
    double wu;  // input weight R_a of this lane's input slot a = L & 3 (kept in a register: a select
                // chain at the point of use gets turned into a lookup table in scratch, whose load
                // then sits in the middle of the prefetch queue of every stage)
};
// An opaque copy of the lane indices: addresses formed from it cannot be hoisted above this point (a kernel that calls several
// sweeps inside an iteration loop otherwise keeps every sweep's loop-invariant addresses and selects live over the whole loop).
__device__ __forceinline__ Lane lane_opaque(const Lane& t) {
    Lane o = t;
    asm volatile("" : "+v"(o.L), "+v"(o.q), "+v"(o.wave), "+v"(o.inst));
    return o;
}
__device__ __forceinline__ Lane lane_id(const Params& P) {
    Lane t;
    t.L = threadIdx.x & 15;
    t.row = threadIdx.x >> 4;
    t.q = t.row;
    t.wave = blockIdx.x;
    t.inst = t.wave * 4 + t.q;
    t.valid = t.inst < P.B;
    t.wu = lane_wu(P, t.L & 3);
    return t;
}
// Row r of this wavefront works on an arbitrary instance (compacted interior-point waves);
// rows without work are parked on the spare workspace block NW (never read by anyone else).
__device__ __forceinline__ Lane lane_indirect(const Params& P, int inst, bool valid) {
    Lane t;
    t.L = threadIdx.x & 15;
    t.row = threadIdx.x >> 4;
    t.inst = valid ? inst : P.NW * 4 + t.row;
    t.wave = t.inst >> 2;
    t.q = t.inst & 3;
    t.valid = valid;
    t.wu = lane_wu(P, t.L & 3);
    return t;
}
// Workspace pointers live inside the by-value Params struct, where clang cannot infer the
// address space: gm() re-types them as global (address_space(1)) so that loads / stores are
// global_* instead of flat_*.
typedef __attribute__((address_space(1))) double gdouble;
typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) dbl2 gdbl2;
typedef __attribute__((address_space(1))) int gint;
__device__ __forceinline__ gdouble* gm(double* p) { return (gdouble*)(unsigned long long)p; }
__device__ __forceinline__ const gdouble* gm(const double* p) { return (const gdouble*)(unsigned long long)p; }
__device__ __forceinline__ gint* gm(int* p) { return (gint*)(unsigned long long)p; }
typedef __attribute__((address_space(1))) unsigned char gbyte;
__device__ __forceinline__ gbyte* gm(unsigned char* p) { return (gbyte*)(unsigned long long)p; }
// (wave, stage) block of a field
__device__ __forceinline__ gdouble* blk(double* f, const Lane& t, int nst, int k, int sz) {
    return gm(f) + ((size_t)t.wave * nst + k) * sz;
}
__device__ __forceinline__ gdouble* blkab(const Params& P, double* f, const Lane& t, int k, int sz) {
    return gm(f) + abidx(P, t.wave, k) * sz;
}
// hides a value from common-subexpression elimination (keeps broadcast temporaries short-lived)
__device__ __forceinline__ void opaque(double& x) { asm volatile("" : "+v"(x)); }
// a value produced by one of the asm primitives of cfnmpc_dpp.hpp is about to be read through
// DPP by compiler-generated code (bc<>): give it the two wait states hipcc cannot know about
__device__ __forceinline__ void settle(double& x) { asm volatile("s_nop 1" : "+v"(x)); }
// makes a value live in every lane at this point (stops the compiler from sinking the load that
// produced it into a divergent branch)
__device__ __forceinline__ void pin(double& x) { asm volatile("" : "+v"(x)); }
// 4-vectors (inputs, feed-forward terms, input steps, interior-point state, per-stage boxes) in P's own layout (Params.v4b):
// wave-blocked [wave][stage][inst & 3][4] for the home arrays, instance-major [inst][stage][4] for the compact copies
__device__ __forceinline__ size_t i4(const Params& P, const Lane& t, int k, int a) {
    // one address expression for both layouts (the selects are on loop-invariant 32-bit values: a second 64-bit expression
    // behind a select costs k_factor its two-waves-per-SIMD register budget)
    const int rb = P.v4b ? t.wave : t.inst, s4 = P.v4b ? 16 : 4, q4 = P.v4b ? t.q * 4 : 0;
    return ((size_t)rb * P.N + k) * s4 + q4 + a;
}

// Input box of element idx (instance-major 4-vector index in P's own indexing): the scalar box of cfnmpc_set_box,
// or -- SBOX, kernels instantiated for cfnmpc_set_box_stages -- the per-stage, per-input arrays (acados' "lbu" /
// "ubu" on individual stages, acados_mpc.cpp:605-608).
template <bool SBOX>
__device__ __forceinline__ void box_at(const Params& P, const size_t idx, double& lo, double& hi) {
    if (SBOX) { lo = gm(P.lbs)[idx]; hi = gm(P.ubs)[idx]; }
    else { lo = P.u_min; hi = P.u_max; }
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
// Unmasked variants: for operands that are consumed ONLY as DPP broadcast sources (always read
// from lanes inside the stored range) the out-of-range lanes may hold anything -- no selects.
__device__ __forceinline__ void ld_ar_raw(const gdouble* b, const Lane& t, double (&ar)[10]) {
    SFOR(s, 0, 10, { ar[s] = b[4 * ar_pre(s) + t.q * ar_n(s) + imin(t.L, ar_n(s) - 1)]; });
}
__device__ __forceinline__ void ld_rows4_raw(const gdouble* b, const Lane& t, double (&r)[4]) {
    SFOR(a, 0, 4, { r[a] = b[(a * 4 + t.q) * 13 + imin(t.L, 12)]; });
}
__device__ __forceinline__ void ld_cols4_raw(const gdouble* b, const Lane& t, double (&c)[13]) {
    SFOR(l, 0, 13, { c[l] = b[(l * 4 + t.q) * 4 + (t.L & 3)]; });
}
// Selects of the masked loaders as functions of their own: a stage that is PREFETCHED is loaded raw and masked where it is used --
// a select placed right behind the load (the compiler does, when registers are tight) waits for the load it belongs to and, vector-
// memory operations retiring in issue order, for everything requested before it: the whole prefetch queue.
__device__ __forceinline__ double ld13_raw(const gdouble* b, const Lane& t) { return b[t.q * 13 + imin(t.L, 12)]; }
__device__ __forceinline__ double mask13(const Lane& t, double v) { return t.L < 13 ? v : 0.0; }
__device__ __forceinline__ void mask_ar(const Lane& t, const double (&in)[10], double (&ar)[10]) {
    SFOR(s, 0, 10, { ar[s] = t.L < ar_n(s) ? in[s] : 0.0; });
}
__device__ __forceinline__ void mask_rows4(const Lane& t, const double (&in)[4], double (&r)[4]) {
    SFOR(a, 0, 4, { r[a] = t.L < 13 ? in[a] : 0.0; });
}
__device__ __forceinline__ void mask_cols4(const Lane& t, const double (&in)[13], double (&c)[13]) {
    SFOR(l, 0, 13, { c[l] = t.L < 4 ? in[l] : 0.0; });
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
// Riccati sweeps
// =============================================================================================
// 1/sqrt(s) to full double accuracy: hardware estimate + two Newton steps (no IEEE div / sqrt
// sequences in the per-stage critical path)
__device__ __forceinline__ double rsqrt_nr(double s) {
    double y = __builtin_amdgcn_rsq(s);
    const double hs = 0.5 * s;
    y = y * (1.5 - hs * y * y);
    y = y * (1.5 - hs * y * y);
    return y;
}
// 1/x to full double accuracy: hardware estimate + two Newton steps (instead of the IEEE
// division sequence; the interior-point iteration is self-correcting at rounding level)
__device__ __forceinline__ double rcp_nr(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = r * (2.0 - x * r);
    r = r * (2.0 - x * r);
    return r;
}
// Symmetric positive definite 4x4 (packed upper) -> inverse (packed upper) by Cholesky, in stages,
// so that the caller can place independent work between the four pivots (each is a dependent
// chain through v_rsq_f64 and two Newton steps; with one or two waves per SIMD nothing else hides
// that latency).  c.ok = false if a pivot is not positive.
struct Chol4 {
    double Lm[4][4], Li[4][4];
    bool ok;
};
template <int J>
__device__ __forceinline__ void chol4_pivot(const double (&S)[10], Chol4& c) {
    double s = S[s4(J, J)];
    SFOR(k, 0, J, { s -= c.Lm[J][k] * c.Lm[J][k]; });
    c.ok = (J == 0 ? true : c.ok) && (s > 0.0);
    const double inv = rsqrt_nr(s);   // 1 / L_jj
    c.Lm[J][J] = s * inv;
    c.Li[J][J] = inv;
    SFOR(i, J + 1, 4, {
        double tt = S[s4(i, J)];
        SFOR(k, 0, J, { tt -= c.Lm[i][k] * c.Lm[J][k]; });
        c.Lm[i][J] = tt * inv;
    });
}
__device__ __forceinline__ void chol4_finish(Chol4& c, double (&Si)[10]) {
    SFOR(j, 0, 4, {
        SFOR(i, j + 1, 4, {
            double tt = 0;
            SFOR(k, j, i, { tt -= c.Lm[i][k] * c.Li[k][j]; });
            c.Li[i][j] = tt * c.Li[i][i];
        });
    });
    SFOR(i, 0, 4, {
        SFOR(j, i, 4, {
            double tt = 0;
            SFOR(k, j, 4, { tt += c.Li[k][i] * c.Li[k][j]; });
            Si[s4(i, j)] = tt;
        });
    });
}

// Everything one factorisation stage reads from HBM (so that the caller can prefetch it).
template <bool ABSOLUTE>
struct StageIn {
    double ar[10], br[4];
    double Rh, g;            // lanes a < 4: input Hessian diagonal / gradient element a
    double bv, qv;           // ABSOLUTE: b_k[i] and q_k[i] = Q_i (x_k[i] - yref_k[i]) in lane i
};
// wq: this lane's state weight Q_i (lane i < 13), computed once per kernel
template <bool ABSOLUTE>
__device__ __forceinline__ void load_stage(const Params& P, const Lane& t, const int k, const double wq,
                                           StageIn<ABSOLUTE>& in) {
    ld_ar_raw(blkab(P, P.AR, t, k, SZ_A), t, in.ar);
    ld_rows4_raw(blkab(P, P.BR, t, k, SZ_B), t, in.br);
    const int a = t.L & 3;
    if (ABSOLUTE) {
        const double uk = blk(P.uit, t, P.N, k, SZ_V4)[t.q * 4 + a];   // (start solve: home arrays, wave-blocked)
        const gdouble* yb = blk(P.yref, t, P.N, k, SZ_Y);
        const double yr = yb[t.q * 17 + 13 + a];
        const double wa = t.wu;
        in.Rh = wa;                 // read in lanes a < 4 only
        in.g = wa * (uk - yr);
        in.bv = blkab(P, P.b, t, k, SZ_V13)[t.q * 13 + imin(t.L, 12)];
        const double xk = blk(P.xit, t, P.N + 1, k, SZ_V13)[t.q * 13 + imin(t.L, 12)];
        const double yk = yb[t.q * 17 + imin(t.L, 12)];
        in.qv = wq * (xk - yk);     // q_k[i] in lane i < 13
    } else {
        in.Rh = gm(P.Rh)[i4(P, t, k, a)];
        in.g = gm(P.g)[i4(P, t, k, a)];
        in.bv = 0.0;
        in.qv = 0.0;
    }
}

// LDS tile of the W transpose, one per row group: lane L writes its row W[L][0..12] as one
// contiguous run (16-byte stores, row stride WT_ROW = 14 doubles: the 13 runs of a row group fall
// on disjoint banks) and reads column l as wt[l * WT_ROW + L] (consecutive lanes = consecutive
// banks).  WT_TILE = 204 doubles staggers the four row groups of a wave by 24 banks, so that their
// 26-bank runs collide two-fold at most (they collided four-fold with a 224-double tile).
constexpr int WT_ROW = 14, WT_TILE = 204;
static_assert(WT_TILE >= 13 * WT_ROW && WT_TILE % 2 == 0, "W transpose tile");
// One stage of the augmented backward recursion.
//   Pa[13]: lanes 0..12 row i of P_{k+1}; lane 13 the affine row p_{k+1}' -- on exit the same for
//           stage k.  ABSOLUTE: start solve with the QP's own affine terms (q_k, b_k, r_k);
//   otherwise R^ and g come from the interior-point state (homogeneous Newton system).
//   wt: LDS [13*17] (transpose of W), sb: LDS [4*16] (columns of B for lanes 0..3).
//   ZL: the stage's outputs go to the instance-contiguous compact store of the level-synchronous
//   active-set passes (t.inst = compact slot; layouts at zrow() below); act = false: compute only,
//   store nothing (a row of a pass wave that has not joined the backward sweep yet).
//   QTAB (fused start solve, cfnmpc_linfactor.hip): the diagonal weights enter through an LDS table -- qtab[L * QT_ROW + j]
//   = Q_j for L == j < 13, qtab[L * QT_ROW + 13 + c] = R_c for L == c < 4, 0 elsewhere (qtab_fill) -- instead of 17
//   per-lane selects, which the compiler hoists out of the stage loop into 34 registers that kernel does not have.
constexpr int QT_ROW = 18;
__device__ __forceinline__ void qtab_fill(const Params& P, double* qtab) {   // one wavefront: lanes 0..15 fill their rows
    const int L = threadIdx.x;
    if (L < 16) {
        SFOR(j, 0, 13, { qtab[L * QT_ROW + j] = (L == j) ? P.W[ext_of(j)] : 0.0; });
        SFOR(c, 0, 4, { qtab[L * QT_ROW + 13 + c] = (L == c) ? P.W[13 + c] : 0.0; });
        qtab[L * QT_ROW + 17] = 0.0;
    }
}
template <bool ABSOLUTE, bool AS = false, bool ZL = false, bool QTAB = false>
__device__ __forceinline__ bool factor_stage(const Params& P, const Lane& t, const int k, double (&Pa)[13],
                                             const StageIn<ABSOLUTE>& in, const double wq, const double is13,
                                             double* wt, double* sb, const bool act = true, const double* qtab = nullptr) {
    const double(&ar)[10] = in.ar;
    const double(&br)[4] = in.br;
    if (ABSOLUTE) {
        // hb' = p' + (P b)' in lane 13
        double pb = 0.0;
        dotbc<13, 0>(pb, Pa, in.bv);          // lanes 0..12: (P b)[i]
        rank1bc<13>(Pa, is13, pb);   // lane 13: p'[j] += (P b)[j]
    }
    // (1) W = Pa A (row form, instruction-level sparsity of A), (2) V = Pa B
    double W[13], V[4];
    SFOR(j, 0, 3, { W[j] = Pa[j]; });
    SFOR(j, 3, 13, { W[j] = 0.0; });
    // (three / four accumulator chains side by side: a wave alone on its SIMD issues dependent FP64 operations at half rate)
    dot3bc<6>(W[3], W[4], W[5], Pa, ar[0], ar[1], ar[2]);
    dot4bc<10>(W[6], W[7], W[8], W[9], Pa, ar[3], ar[4], ar[5], ar[6]);
    dot3bc<13>(W[10], W[11], W[12], Pa, ar[7], ar[8], ar[9]);
    // (3) Wt = transpose of W over lanes 0..12 through the LDS tile (lane 13 keeps the affine row);
    //     the same round trip hands the columns of B to lanes 0..3.  The tile is written BEFORE V
    //     is formed and S is formed BEFORE the transposed rows are used, so that both LDS
    //     latencies sit behind 52 broadcast FMAs each.
    double Wt[13];
    __syncthreads();
    if (t.L < 13) {
        SFOR(j, 0, 13, { wt[t.L * WT_ROW + j] = W[j]; });
        SFOR(a, 0, 4, { sb[a * 16 + t.L] = br[a]; });
    }
    SFOR(a, 0, 4, { V[a] = 0.0; });
    dot4bc<13>(V[0], V[1], V[2], V[3], Pa, br[0], br[1], br[2], br[3]);
    __syncthreads();
    double bcl[13];   // lanes >= 4 compute don't-care rows of S (never broadcast)
    SFOR(l, 0, 13, { bcl[l] = sb[(t.L & 3) * 16 + l]; });
    // (all 13 reads issued unconditionally, then pinned: otherwise the compiler sinks every
    //  read into its own branch on "lane != 13")
    SFOR(l, 0, 13, { Wt[l] = wt[l * WT_ROW + imin(t.L, 12)]; });
    // (4) S = R^ + B'V in lanes a < 4, replicated; every lane inverts it redundantly (4x4 Cholesky),
    //     one pivot at a time BETWEEN the blocks of (5) and (6), which hide the pivots' latency
    double Srow[4];
    if (QTAB && ABSOLUTE && !AS) SFOR(c, 0, 4, { Srow[c] = qtab[t.L * QT_ROW + 13 + c]; });   // (R constant: start solve only)
    else SFOR(c, 0, 4, { Srow[c] = (t.L == c) ? in.Rh : 0.0; });
    dot4bc<13>(Srow[0], Srow[1], Srow[2], Srow[3], bcl, V[0], V[1], V[2], V[3]);
    SFOR(c, 0, 4, { settle(Srow[c]); });
    if (AS && t.L < 4 && act) {
        gdouble* sr = ZL ? gm(P.cS) + ((size_t)t.inst * P.N + k) * 16 + t.L : blk(P.cS, t, P.N, k, SZ_S4) + t.q * 4 + t.L;
        SFOR(c, 0, 4, { sr[c * (ZL ? 4 : 16)] = Srow[c]; });
    }
    double S[10], Si[10];
    SFOR(a, 0, 4, { SFOR(c, a, 4, { S[s4(a, c)] = bc<a>(Srow[c]); }); });
    SFOR(l, 0, 13, { pin(Wt[l]); });
    SFOR(l, 0, 13, { Wt[l] = t.L == 13 ? Pa[l] : Wt[l]; });   // lanes 14, 15: don't-care (never broadcast)
    Chol4 ch;
    chol4_pivot<0>(S, ch);
    // (5) M = Q + Wt A  (lane 13: q_k' + hb'A)
    double M[13];
    if (QTAB) SFOR(j, 0, 13, { M[j] = qtab[t.L * QT_ROW + j]; });
    else SFOR(j, 0, 13, { M[j] = (t.L == j) ? wq : 0.0; });
    if (ABSOLUTE) rank1bc<13>(M, is13, in.qv);   // lane 13: += q_k[j]
    SFOR(j, 0, 3, { M[j] += Wt[j]; });
    dot3bc<6>(M[3], M[4], M[5], Wt, ar[0], ar[1], ar[2]);
    chol4_pivot<1>(S, ch);
    dot4bc<10>(M[6], M[7], M[8], M[9], Wt, ar[3], ar[4], ar[5], ar[6]);
    chol4_pivot<2>(S, ch);
    dot3bc<13>(M[10], M[11], M[12], Wt, ar[7], ar[8], ar[9]);
    chol4_pivot<3>(S, ch);
    // (6) G' = Wt B ; lane 13: rho = g + B'hb
    double Gp[4];
    SFOR(a, 0, 4, { Gp[a] = 0.0; });
    dot4bc<13>(Gp[0], Gp[1], Gp[2], Gp[3], Wt, br[0], br[1], br[2], br[3]);
    rank1bc<4>(Gp, is13, in.g);   // lane 13: += g[a]
    if (AS) {
        // active-set solve: the forward sweep evaluates the multipliers of the fixed inputs from the
        // stage's own blocks, B'pi_{k+1} = G dx_k + (B'PB) du_free + rho -- keep G (gain layout),
        // rho (lane 13) and the rows of S (off-diagonal entries = B'PB, untouched by the fixing weight)
        gdouble* gr = ZL ? gm(P.cGR) + ((size_t)t.inst * P.N + k) * 52 + imin(t.L, 12) * 4
                         : blk(P.cGR, t, P.N, k, SZ_K) + (imin(t.L, 12) * 4 + t.q) * 4;
        gdouble* dst = t.L == 13 ? gm(P.crho) + i4(P, t, k, 0) : gr;
        if (t.L < 14 && act) SFOR(a, 0, 4, { dst[a] = Gp[a]; });
    }
    chol4_finish(ch, Si);
    const bool ok = ch.ok;
    // (7) K' = G' Sinv  (lane 13: feed-forward d)
    double Kp[4], nGp[4];
    SFOR(a, 0, 4, {
        double s = 0.0;
        SFOR(c, 0, 4, { s += Gp[c] * Si[s4(c, a)]; });
        Kp[a] = s;
        nGp[a] = -Gp[a];
    });
    // (8) P <- M - G' K  (lane 13: p' <- M_13 - rho' K)
    SFOR(j, 0, 13, { Pa[j] = M[j]; });
    upd4bc<0, 4>(Pa, Kp, nGp);
    upd4bc<4, 4>(Pa, Kp, nGp);
    upd4bc<8, 4>(Pa, Kp, nGp);
    upd4bc<12, 1>(Pa, Kp, nGp);
    // (9) stores: gain in "lane a holds K[a][.]" form, Sinv, feed-forward
    {
        // lanes 0..12 store their column of the gain, lane 13 the feed-forward: one masked
        // region with per-lane addresses
        gdouble* kr = ZL ? gm(P.KR) + ((size_t)t.inst * P.N + k) * 52 + imin(t.L, 12) * 4
                         : blk(P.KR, t, P.N, k, SZ_K) + (imin(t.L, 12) * 4 + t.q) * 4;
        gdouble* dst = t.L == 13 ? ((ABSOLUTE && !AS) ? blk(P.d, t, P.N, k, SZ_V4) + t.q * 4   // (start solve: home array, wave-blocked)
                                                      : gm(P.d) + i4(P, t, k, 0)) : kr;
        if (t.L < 14 && act) SFOR(a, 0, 4, { dst[a] = Kp[a]; });
        if (!ABSOLUTE && t.L == 0) {  // only the corrector of the interior-point iteration reads it
            gdouble* sv = blk(P.Sinv, t, P.N, k, SZ_S);
            SFOR(e, 0, 10, { sv[t.q * 10 + e] = Si[e]; });
        }
    }
    return ok;
}

}  // namespace cfn
