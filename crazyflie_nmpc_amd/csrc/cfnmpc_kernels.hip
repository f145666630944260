// cfnmpc_kernels.hip -- HIP kernels of the batched Crazyflie SQP-RTI step (gfx950, FP64).
//
// Two mappings, chosen per kernel by what the kernel carries from stage to stage:
//   * row groups (cfnmpc_ws.hpp): one NMPC instance per 16-lane DPP row, four instances per
//     wavefront, lane i = row i of every 13-row object -- for the Riccati recursions, which carry
//     the 13 x 13 cost-to-go (k_factor, k_as, k_ipm_rest);
//   * lane per instance: 64 instances per wavefront, all model arithmetic in registers, 13-vectors
//     through LDS tiles -- for the kernels that carry at most a 13-vector (k_linearise, k_forward).
// One wavefront per workgroup; instances never communicate.
//
// Kernels, one RTI step = eight launches on one stream (large fleets; small ones: see launch_qp_*):
//   k_linearise : RK4 + forward sensitivities per shooting interval (the role of acados sim_erk +
//                 CasADi forw_vde, acados_mpc.cpp:84), written in the row-distributed A / B form.
//   k_factor    : start solve, backward: augmented Riccati factorisation of the unconstrained QP
//                 over all N stages, next stage software-prefetched (2 waves/SIMD).
//   k_forward   : start solve, forward, MATRIX-FREE: dx+ = A dx + B du + b is evaluated as the
//                 directional derivative of the RK4 map (one forward-mode pass) instead of reading
//                 A and B; unconstrained inputs, feasibility / active-horizon decision; the candidate
//                 is written straight into the second iterate buffer, which IS the full RTI step of
//                 the instances whose unconstrained minimiser respects the input box (the host
//                 swaps the two iterate buffers after the step).
//   k_compact   : list of the instances that need the interior-point method, by head class.
//   k_scatter   : places every constrained instance in the list.
//   k_as        : those instances only, on a compact copy of their head stages: primal-dual
//                 active-set solves (one Riccati factorisation with the active inputs fixed and one
//                 forward sweep that also evaluates the multipliers and re-classifies) until the
//                 active set is stationary = exact QP solution; roll-out into the second iterate
//                 buffer with tail verification (acados_solve() epilogue, acados_mpc.cpp:611-616).
//   k_ipm_list  : compacts the rows k_as left (large fleets), four per wave for
//   k_ipm_rest  : rows k_as left (no stationary set within 12 solves, tail check failed, or too far outside the
//                 box to try): Mehrotra predictor-corrector over stage-wise Riccati sweeps in delta form, clipped
//                 start for rows far outside the box (HPIPM's role, generate_c_code.py:140).  k_ipm = the same
//                 without k_as (cfnmpc_opts.active_set = 0).
//   k_as_solves / k_ascommit / k_as_retry, k_asf / k_asw / k_asp : scheduling variants of the active-set phase
//                 (cfnmpc_opts.as_passes; solves + commit kernel is the default of small fleets).
//   k_linearise_list : cfnmpc_opts.overlap_linearise only -- re-linearises the interior-point
//                 instances after the early pass that ran beside k_ipm.
//   k_sim / k_estimate : RK4 plant step / predictor (acados_estimator.cpp:573-593).
//   k_put / k_get / k_init_iterate / k_windows : layout glue and reference windows for the C-ABI.
#include <hip/hip_runtime.h>
#include <cstdio>

#include "cfnmpc_rg.hpp"

namespace cfn {

// =============================================================================================
// linearisation
// =============================================================================================
// One sensitivity column through the four RK stages (HQ / HW: its q- / w-part can be non-zero;
// IS_U: input column, driven by df/du).  Column-type sparsity is exploited at compile time
// because all 64 lanes (= 64 instances) integrate the SAME column.
template <bool HQ, bool HW, bool IS_U>
__device__ __forceinline__ void sens_column(const JacPoint (&J)[4], const double* __restrict__ u, int c, double h,
                                            double* __restrict__ col) {
    double s0[13], s[13], k1[13], k2[13], k3[13], k4[13], ju[4] = {0, 0, 0, 0};
    SFOR(i, 0, 13, { s0[i] = (!IS_U && i == c) ? 1.0 : 0.0; });
    if (IS_U) {
        const double uc = 2.0 * (c == 0 ? u[0] : (c == 1 ? u[1] : (c == 2 ? u[2] : u[3])));
        const double sa = (c < 2) ? 1.0 : -1.0;             // w1 w2 | -w3 -w4
        const double sb = (c == 0 || c == 3) ? 1.0 : -1.0;  // w1 -w2 -w3 w4
        const double sc_ = (c == 0 || c == 2) ? 1.0 : -1.0; // w1 -w2 w3 -w4
        ju[0] = KT * uc; ju[1] = KA * sa * uc; ju[2] = KB * sb * uc; ju[3] = KC * sc_ * uc;
    }
    jvp<HQ, HW>(J[0], s0, k1);
    SFOR(i, 0, 4, { k1[9 + i] += ju[i]; });
    SFOR(i, 0, 13, { s[i] = s0[i] + 0.5 * h * k1[i]; });
    jvp<HQ, HW>(J[1], s, k2);
    SFOR(i, 0, 4, { k2[9 + i] += ju[i]; });
    SFOR(i, 0, 13, { s[i] = s0[i] + 0.5 * h * k2[i]; });
    jvp<HQ, HW>(J[2], s, k3);
    SFOR(i, 0, 4, { k3[9 + i] += ju[i]; });
    SFOR(i, 0, 13, { s[i] = s0[i] + h * k3[i]; });
    jvp<HQ, HW>(J[3], s, k4);
    SFOR(i, 0, 4, { k4[9 + i] += ju[i]; });
    SFOR(i, 0, 13, { col[i] = s0[i] + (h / 6.0) * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]); });
}

// Lane-per-instance (work-efficient: nothing is computed twice); one workgroup = one wavefront
// = 64 instances, blockIdx.y = a chunk of the (mutually independent) shooting intervals.  All
// traffic to the wave-blocked layout goes through LDS tiles so that every global load / store
// instruction covers contiguous runs (13 / ar_n doubles per instance).
//   GATHER = false: instances 64 g .. 64 g + 63 (16 consecutive workspace blocks);
//   GATHER = true : the instances P.ilist[64 g ..] (the interior-point instances of this step,
//                   whose iterate only became final after the early pass over everybody).
// Lanes without an instance work on the spare workspace block NW (finite data, never read).
__host__ __device__ constexpr unsigned div_magic(int ns) { return (65536u + (unsigned)ns - 1u) / (unsigned)ns; }
__host__ __device__ constexpr bool div_ok(int ns, unsigned n = 0) {   // exact for e < n (default: 64 * ns)
    for (unsigned e = 0; e < (n ? n : 64u * (unsigned)ns); e++)
        if (((e * div_magic(ns)) >> 16) != e / (unsigned)ns || e * div_magic(ns) >= (1u << 24)) return false;
    return true;
}
static_assert(div_ok(13) && div_ok(10) && div_ok(6), "k_linearise: e / NS by multiply-shift");
// the same with a 20-bit shift (16-byte pieces of a group tile -> block: e < 1664, ns = pieces per block)
__host__ __device__ constexpr unsigned div_magic20(int ns) { return ((1u << 20) + (unsigned)ns - 1u) / (unsigned)ns; }
__host__ __device__ constexpr bool div_ok20(int ns, unsigned n) {
    for (unsigned e = 0; e < n; e++)
        if ((((unsigned long long)e * div_magic20(ns)) >> 20) != e / (unsigned)ns || (unsigned long long)e * div_magic20(ns) >= (1ull << 32) ||
            div_magic20(ns) >= (1u << 24) || e >= (1u << 24))
            return false;
    return true;
}
static_assert(div_ok(52, 64 * 13), "k_forward: e / 52 by multiply-shift");
//   CSTORE (with GATHER): the results go to the COMPACT store of the constrained-QP kernels (P.cAR, P.cBR, P.cbv; list
//                   slot c owns row c & 3 of compact block c >> 2, i.e. 64 consecutive slots = 16 consecutive blocks,
//                   written in the same contiguous runs as the home blocks) -- the fused start solve never stores
//                   (A, B, b), so the instances whose QP needs them again get them here (k_linearise_clist);
//                   which = 0: P.ilist (count P.nipm[0]), 1: P.ilist2 (P.nipm[NI_LISTED], the interior-point fall-back rows).
template <bool GATHER, bool CSTORE = false>
__device__ __forceinline__ void linearise_body(const Params& P, double* sx, double (*sc)[64 * 13], int* sinst,
                                               const int which = 0) {
    constexpr bool PAIRS = !GATHER;   // 16-byte stores of whole column groups (below)
    const int tid = threadIdx.x;
    const double h = P.dt;
    const int N = P.N;
    {
        const int li = blockIdx.x * 64 + tid;
        int inst;
        if (GATHER) inst = li < gm(P.nipm)[which ? NI_LISTED : 0] ? gm(which ? P.ilist2 : P.ilist)[li] : P.NW * 4 + (tid & 3);
        else inst = li < P.NW * 4 ? li : P.NW * 4 + (tid & 3);
        sinst[tid] = inst;
    }
    __syncthreads();
    const int inst = sinst[tid];
#ifdef CFN_DEV   // (P.lin_k1 > 0: only the shooting intervals [lin_k0, lin_k1) -- the stage-chunked hand-over experiment)
    const int ka = P.lin_k1 > 0 ? P.lin_k0 : 0, kb = P.lin_k1 > 0 ? P.lin_k1 : N;
#else
    const int ka = 0, kb = N;
#endif
    const int per = (kb - ka + gridDim.y - 1) / gridDim.y;
    const int k0 = ka + blockIdx.y * per, k1 = imin(kb, k0 + per);
    if (k0 >= k1) return;
    // CSTORE: list slots of this group that own a row (the lanes behind the list's end linearise the spare instance; their
    // results must not land in real compact slots -- for which = 1 those belong to rows of the FIRST list)
    const int n_live = CSTORE ? gm(P.nipm)[which ? NI_LISTED : 0] - (int)blockIdx.x * 64 : 64;

    // address of element (local instance li, lane i) of a field with `stages` stages per block, SZ
    // doubles per (block, stage), NS lanes per instance starting at `pre4` inside the block.
    // Consecutive instances: a wave-uniform 64-bit base (scalar registers) + a 32-bit byte offset
    // per lane (one multiply-add instead of four 64-bit operations per element; 16 blocks x stages x
    // SZ x 8 bytes < 4 GB for any admissible horizon), i.e. the saddr form of global_load / store.
    auto el = [&](double* field, int li, int i, int stages, int k, int SZ, int pre4, int NS, bool home = true) -> gdouble* {
        if (GATHER && (home || !CSTORE)) {
            const int in = sinst[li];
            // (loads: the iterate, [block][stage]; stores of the list kernel without CSTORE: the home A, B, b, grouped by 16 blocks)
            const size_t bs = home ? (size_t)(in >> 2) * stages + k : ((size_t)(in >> 6) * stages + k) * 16 + ((in >> 2) & 15);
            return gm(field) + bs * SZ + pre4 + (in & 3) * NS + i;
        }
        const int w0 = (int)blockIdx.x * 16;
        // (24-bit multiply: full rate, the 32-bit one takes four issue slots; block index <= 16, stages * SZ < 2^24 for N <= 4096)
        const unsigned off = (__umul24((unsigned)imin(li >> 2, P.NW - w0), (unsigned)(stages * SZ)) + (unsigned)((li & 3) * NS + i)) * 8u;
        const char* base = (const char*)(gm(field) + ((size_t)w0 * stages + k) * SZ + pre4);
        return (gdouble*)(base + off);
    };
    // Trip counts are compile-time (64 instances x NS lanes) and the loops fully unrolled, so that
    // all loads / LDS reads of a transfer are in flight together.
    auto issue_x = [&](int k, int tl, double (&xr)[13]) {  // stage k of xit -> registers (no wait)
        SFOR(r, 0, 13, {
            const int e = tl + 64 * r;
            const int li = (int)(__umul24((unsigned)e, div_magic(13)) >> 16), i = e - li * 13;
            xr[r] = *el(P.xit, li, i, N + 1, k, SZ_V13, 0, 13);
        });
    };
    auto land_x = [&](double* tile, const double (&xr)[13]) {
        SFOR(r, 0, 13, { tile[tid + 64 * r] = xr[r]; });
    };
    auto issue_u = [&](int k, double (&u)[4]) {
        const gdouble* up = gm(P.uit) + (P.v4b ? (((size_t)(inst >> 2) * N + k) * 4 + (inst & 3)) * 4 : ((size_t)inst * N + k) * 4);
        SFOR(a, 0, 4, { u[a] = up[a]; });
    };
    // Vector-memory operations complete in issue order on this part (one counter for loads and stores): waiting for a load
    // waits for every store issued before it.  So the NEXT stage's inputs are requested at the top of a stage and used one
    // stage later, when the 82 (162) stores issued in between have pushed them out of the 63-deep window -- the stage loop
    // never waits for its own stores to drain (requested at the end of the stage, as before round 5, every stage ended with
    // the wave waiting for its A stores: 1.15 -> 1.00 ms at 65 536 instances together with the 16-byte stores below).
    // Two state tiles (round 6): sxa = x_k, sxb = x_{k+1}.  x_{k+2} is requested at the top of stage k and LANDED in the middle
    // of the same stage (before the rate columns, into the tile x_k was read from): by then 71 younger stores have pushed it out
    // of the window, so the wait is free -- and its 26 registers are free during the rate and input columns, where the four
    // Jacobian points + a column + its temporaries need the whole file (before: carried to the next stage's top, 52 B of
    // scratch with three spill / reload pairs inside the stage loop).
    //   on entry to stage k: sxa = x_k, sxb = x_{k+1} (tiles), un = u_k as loaded (maybe still in flight)
    double xr[13], un[4], xn[13];
    double *sxa = sx, *sxb = sx + 64 * 13;
    {
        issue_x(k0, tid, xr);
        land_x(sxa, xr);
        issue_x(k0 + 1, tid, xr);
        land_x(sxb, xr);
        __syncthreads();
        issue_u(k0, un);
        // (waited for HERE, so that the waits the compiler places at the loop's top are sized for the path around the loop --
        //  82 younger stores, i.e. none -- and not for this entry)
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    }
    for (int k = k0; k < k1; k++) {
        double x[13], u[4];
        SFOR(e, 0, 13, { x[e] = sxa[tid * 13 + int_of(e)]; });
        SFOR(a, 0, 4, { u[a] = un[a]; });
        {
            int tq = tid;
            asm volatile("" : "+v"(tq));
            issue_x(imin(k + 2, N), tq, xr);
            issue_u(imin(k + 1, N - 1), un);
        }
        int tl = tid;  // opaque per-stage copy: keeps the store offsets (82, or 162 in the list kernels) from being hoisted out
        asm volatile("" : "+v"(tl));  // of the stage loop (they would occupy that many registers for its whole length)
        // nominal RK4 (classic tableau, one step per interval)
        double xt[13], k1v[13], k2v[13], k3v[13], k4v[13];
        JacPoint J[4];
        f_expl(x, u, k1v);
        jac_point(x, J[0]);
        SFOR(e, 0, 13, { xt[e] = x[e] + 0.5 * h * k1v[e]; });
        f_expl(xt, u, k2v);
        jac_point(xt, J[1]);
        SFOR(e, 0, 13, { xt[e] = x[e] + 0.5 * h * k2v[e]; });
        f_expl(xt, u, k3v);
        jac_point(xt, J[2]);
        SFOR(e, 0, 13, { xt[e] = x[e] + h * k3v[e]; });
        f_expl(xt, u, k4v);
        jac_point(xt, J[3]);
        SFOR(e, 0, 13, { xn[e] = sxb[tid * 13 + int_of(e)]; });
        // b = Phi - x_{k+1} through the tile (internal order)
        SFOR(r, 0, 13, {
            constexpr int e = ext_of(r);
            const double phi = x[e] + (h / 6.0) * (k1v[e] + 2 * k2v[e] + 2 * k3v[e] + k4v[e]);
            sc[0][tid * 13 + r] = phi - xn[e];
        });
        __syncthreads();
        // e / NS for e < 64 * NS as a 24-bit multiply + shift (verified at compile time: div_ok)
#define CFN_DIV(e, NS) ((int)(__umul24((unsigned)(e), div_magic(NS)) >> 16))
        // PAIRS (the kernel over the whole fleet): a group of columns leaves through ONE tile that mirrors the group's
        // region of the 16 stage blocks -- [block][column][instance of the block][row < NS], CH doubles per block, the
        // order they have in HBM -- so that tile -> field is a linear copy in 16-byte pieces: piece g of the tile
        // (ds_read_b128 at 16 g) goes to block g / (CH / 2), 16-byte piece g % (CH / 2) of the region.  Half the store
        // instructions of the 8-byte form (82 instead of 162 per stage) at twice the bytes each.
        // tile column j of a group: lane base tb(CH, NS) + 4 NS j + row
        double* const scf = &sc[0][0];
#define CFN_TB(CH, NS) ((tid >> 2) * (CH) + (tid & 3) * (NS))
#define CFN_COL(call, j, CH, NS)                                                                       \
    {                                                                                                   \
        call;                                                                                           \
        if (PAIRS) {                                                                                    \
            const int tb = CFN_TB(CH, NS) + 4 * (NS) * j;                                               \
            SFOR(r, 0, NS, { scf[tb + r] = col[ext_of(r)]; });                                          \
        } else {                                                                                        \
            SFOR(r, 0, 13, { sc[j][tid * 13 + r] = col[ext_of(r)]; });                                  \
        }                                                                                               \
    }
        // rows < NS of column tile `ti` -> `field` (SZ doubles per block and stage) at `pre4` (8-byte form: the list kernels)
#define CFN_STORE(field, SZ, ti, NS, pre4)                                                              \
    {                                                                                                   \
        double tv[NS];                                                                                  \
        SFOR(r, 0, NS, {                                                                                \
            const int e = tl + 64 * r;                                                                  \
            const int li = CFN_DIV(e, NS), i = e - li * (NS);                                          \
            tv[r] = sc[ti][li * 13 + i];                                                                \
        });                                                                                             \
        SFOR(r, 0, NS, {                                                                                \
            const int e = tl + 64 * r;                                                                  \
            const int li = CFN_DIV(e, NS), i = e - li * (NS);                                          \
            if (!CSTORE || li < n_live) *el(field, li, i, N, k, SZ, pre4, NS, false) = tv[r];            \
        });                                                                                             \
    }
        // the group tile (CH doubles per block) -> `field` at PRE inside every block's stage (16-byte form)
#define CFN_STORE2(field, SZ, CH, PRE)                                                                  \
    {                                                                                                   \
        constexpr int HALF = (CH) / 2, NP = 16 * HALF, NI = (NP + 63) / 64;                             \
        static_assert(div_ok20(HALF, 64 * NI) && 128 * NI <= 4 * 64 * 13, "piece -> block by multiply-shift; tile size"); \
        dbl2 tv[NI];                                                                                    \
        SFOR(r, 0, NI, { tv[r] = *(const dbl2*)&scf[2 * (tl + 64 * r)]; });                             \
        /* (home layout: the 16 blocks of a group lie side by side per stage, cfnmpc_rg.hpp: abidx; whole groups exist) */ \
        const unsigned sdiff8 = (unsigned)((SZ) - (CH)) * 8u;                                           \
        const char* base = (const char*)(gm(field) + ((size_t)blockIdx.x * N + k) * 16 * (SZ) + (PRE)); \
        SFOR(r, 0, NI, {                                                                                \
            const unsigned g = (unsigned)(tl + 64 * r);                                                 \
            const unsigned bq = __umul24(g, div_magic20(HALF)) >> 20;                                   \
            const unsigned off = __umul24(bq, sdiff8) + 16u * g;                                        \
            if (NP % 64 == 0 || r + 1 < NI || tid < NP % 64) *(gdbl2*)(base + off) = tv[r];             \
        });                                                                                             \
    }
        if (PAIRS) { CFN_STORE2(P.b, SZ_V13, 52, 0); } else { CFN_STORE(CSTORE ? P.cbv : P.b, SZ_V13, 0, 13, 0); }
        double col[13];
        // state columns in internal order: v (internal 3..5 = external 7..9), q (6..9 = 3..6), w (10..12)
        __syncthreads();
        // (runtime loops on purpose: one column at a time keeps the register footprint small)
#pragma unroll 1
        for (int j = 0; j < 3; j++) {  // velocity columns: rows p, v
            CFN_COL((sens_column<false, false, false>(J, u, 7 + j, h, col)), j, 72, 6);
        }
        __syncthreads();
        if (PAIRS) { CFN_STORE2(P.AR, SZ_A, 72, 0); }
        else { SFOR(j, 0, 3, { CFN_STORE(CSTORE ? P.cAR : P.AR, SZ_A, j, ar_n(j), 4 * ar_pre(j)); }); }
        __syncthreads();
#pragma unroll 1
        for (int j = 0; j < 4; j++) {  // quaternion columns: rows p, v, q
            CFN_COL((sens_column<true, false, false>(J, u, 3 + j, h, col)), j, 160, 10);
        }
        __syncthreads();
        if (PAIRS) { CFN_STORE2(P.AR, SZ_A, 160, 4 * ar_pre(3)); }
        else { SFOR(j, 0, 4, { CFN_STORE(CSTORE ? P.cAR : P.AR, SZ_A, j, ar_n(3 + j), 4 * ar_pre(3 + j)); }); }
        // x_{k+2} -> the tile x_k was read from (every lane read it at the stage's top, barriers in between); tiles swap roles
        land_x(sxa, xr);
        { double* t_ = sxa; sxa = sxb; sxb = t_; }
        __syncthreads();
#pragma unroll 1
        for (int j = 0; j < 3; j++) {  // rate columns: all rows
            CFN_COL((sens_column<true, true, false>(J, u, 10 + j, h, col)), j, 156, 13);
        }
        __syncthreads();
        if (PAIRS) { CFN_STORE2(P.AR, SZ_A, 156, 4 * ar_pre(7)); }
        else { SFOR(j, 0, 3, { CFN_STORE(CSTORE ? P.cAR : P.AR, SZ_A, j, ar_n(7 + j), 4 * ar_pre(7 + j)); }); }
        __syncthreads();
#pragma unroll 1
        for (int a = 0; a < 4; a++) {  // input columns: all rows
            CFN_COL((sens_column<true, true, true>(J, u, a, h, col)), a, 208, 13);
        }
        __syncthreads();
        if (PAIRS) { CFN_STORE2(P.BR, SZ_B, 208, 0); }
        else { SFOR(a, 0, 4, { CFN_STORE(CSTORE ? P.cBR : P.BR, SZ_B, a, 13, a * 52); }); }
        __syncthreads();
#undef CFN_STORE2
#undef CFN_TB
#undef CFN_STORE
#undef CFN_DIV
#undef CFN_COL
    }
}
// (Round 4, measured: the same linearisation at TWO waves per SIMD -- RK points kept as q | v | w only (10 instead of 31 doubles per
//  point; a column rebuilds R(q) and forms d(R v) from (q, dq) on the fly), columns leaving in pairs through two LDS tiles, the next
//  state re-fetched instead of carried: 256 registers, 13.6 KB of LDS, parity-green -- needs 9300 instead of 6267 vector instructions
//  per stage (run-time unit vectors and tile indices, rebuilt rotation matrices) and ran 1.69 - 1.75 ms against 1.19 ms at 65 536
//  instances (2048 workgroups of 25 stages), 2.05 ms with one workgroup per 64 instances: the second wave did not raise the VALU
//  occupancy (~54 % either way).  Removed; profiles/r04_linearise_variants.md.)
KALIGN __global__ __launch_bounds__(64) void k_linearise(Params P) {
    __shared__ double sx[2 * 64 * 13];   // two tiles of one 13-vector per instance (internal order): x_k, x_{k+1}
    __shared__ __attribute__((aligned(16))) double sc[4][64 * 13];    // the tile of a group of up to four sensitivity columns
    __shared__ int sinst[64];
    linearise_body<false>(P, sx, sc, sinst);
}
#ifdef CFN_DEV   // (overlapped preparation: development builds only)
__global__ __launch_bounds__(64) void k_linearise_list(Params P) {
    __shared__ double sx[2 * 64 * 13];
    __shared__ double sc[4][64 * 13];
    __shared__ int sinst[64];
    if ((int)blockIdx.x * 64 >= gm(P.nipm)[0]) return;
    linearise_body<true>(P, sx, sc, sinst);
}
#endif
__global__ __launch_bounds__(64) void k_linearise_clist(Params P, int which) {
    __shared__ double sx[2 * 64 * 13];
    __shared__ double sc[4][64 * 13];
    __shared__ int sinst[64];
    if ((int)blockIdx.x * 64 >= gm(P.nipm)[which ? NI_LISTED : 0]) return;
    linearise_body<true, true>(P, sx, sc, sinst, which);
}


// Backward factorisation over stages [0, head), next stage prefetched while the current one is
// computed.  chk >= 0: start from a stored checkpoint of the unconstrained tail (P.Pchk, affine
// row zero), else from the terminal cost.
// klo / park / from_park: only the stages [klo, head) of the sweep, cost-to-go taken from / left in `park`
// ([wave][13][64 lanes]) -- the stage-chunked hand-over experiment (k_factor_chunk); the defaults fold away.
// QTAB: diagonal weights through the LDS table of factor_stage (frees the ~34 registers of their hoisted selects);
// DEEP: three rotating stage buffers -- the loads of stage k - 2 are issued before the arithmetic of stage k (the start
// solve streams 1.75 KB per stage and wave from HBM; with two buffers its waves wait a quarter of their time).
template <bool ABSOLUTE, bool QTAB = false, bool DEEP = false>
__device__ __forceinline__ bool sweep_factor(const Params& P, const Lane& t, const int head, const int chk,
                                             double* wt, double* sb, const int klo = 0, gdouble* park = nullptr,
                                             const bool from_park = false, const double* qtab = nullptr) {
    double Pa[13];
    if (from_park) {
        SFOR(j, 0, 13, { Pa[j] = park[j * 64 + threadIdx.x]; });
    } else if (ABSOLUTE || chk < 0) {
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
    // per-lane constants: state weight of this lane's row, indicator of the affine row
    double wq = 0.0;
    SFOR(j, 0, 13, { if (t.L == j) wq = P.W[ext_of(j)]; });
    const double is13 = t.L == 13 ? 1.0 : 0.0;
    auto after = [&](int k) {
        if (ABSOLUTE) {
            // checkpoints of the unconstrained cost-to-go (matrix part only); one store block with a run-time checkpoint
            // index (six specialised blocks keep six hoisted addresses alive through the whole sweep)
            int cidx = -1;
            SFOR(c, 0, N_CHK, { if (k == chk_stage(c)) cidx = c; });
            if (cidx >= 0) {
                gdouble* pc = gm(P.Pchk) + ((size_t)t.wave * N_CHK + cidx) * SZ_PP;
                SFOR(j, 0, 13, { if (t.L <= j) pc[pchk_col(j) + t.q * (j + 1) + t.L] = Pa[j]; });   // (packed: rows 0..j of column j)
            }
        }
    };
    if constexpr (DEEP) {
        StageIn<ABSOLUTE> b0, b1, b2;
        load_stage<ABSOLUTE>(P, t, head - 1, wq, b0);
        load_stage<ABSOLUTE>(P, t, imax(head - 2, 0), wq, b1);
        int k = head - 1;
        while (k >= klo) {
            load_stage<ABSOLUTE>(P, t, imax(k - 2, 0), wq, b2);
            ok = factor_stage<ABSOLUTE, false, false, QTAB>(P, t, k, Pa, b0, wq, is13, wt, sb, true, qtab) && ok;
            after(k);
            if (--k < klo) break;
            load_stage<ABSOLUTE>(P, t, imax(k - 2, 0), wq, b0);
            ok = factor_stage<ABSOLUTE, false, false, QTAB>(P, t, k, Pa, b1, wq, is13, wt, sb, true, qtab) && ok;
            after(k);
            if (--k < klo) break;
            load_stage<ABSOLUTE>(P, t, imax(k - 2, 0), wq, b1);
            ok = factor_stage<ABSOLUTE, false, false, QTAB>(P, t, k, Pa, b2, wq, is13, wt, sb, true, qtab) && ok;
            after(k);
            --k;
        }
    } else {
    // two stage buffers used alternately (no hand-over copies): while stage k is computed from
    // one, stage k-1 is being loaded into the other
    StageIn<ABSOLUTE> bufA, bufB;
    load_stage<ABSOLUTE>(P, t, head - 1, wq, bufA);
    int k = head - 1;
    while (k >= klo) {
        load_stage<ABSOLUTE>(P, t, imax(k - 1, 0), wq, bufB);
        ok = factor_stage<ABSOLUTE, false, false, QTAB>(P, t, k, Pa, bufA, wq, is13, wt, sb, true, qtab) && ok;
        after(k);
        k--;
        if (k < klo) break;
        load_stage<ABSOLUTE>(P, t, imax(k - 1, 0), wq, bufA);
        ok = factor_stage<ABSOLUTE, false, false, QTAB>(P, t, k, Pa, bufB, wq, is13, wt, sb, true, qtab) && ok;
        after(k);
        k--;
    }
    }
    if (park) SFOR(j, 0, 13, { park[j * 64 + threadIdx.x] = Pa[j]; });
    return ok;
}

// Everything one forward stage reads from HBM: gain, feed-forward, A, B (and b), so that the
// caller can issue the loads of stage k+1 before the arithmetic of stage k (the recursion
// itself only carries the 13-vector x).
template <bool WITH_B>
struct FwdIn {
    double kr[13], ar[10], br[4], d, bv;
};
template <bool WITH_B>
__device__ __forceinline__ void load_fwd(const Params& P, const Lane& t, const int k, FwdIn<WITH_B>& in) {
    // (raw: feedback() / propagate() mask what they read, cfnmpc_rg.hpp: mask_*)
    ld_cols4_raw(blk(P.KR, t, P.N, k, SZ_K), t, in.kr);
    ld_ar_raw(blkab(P, P.AR, t, k, SZ_A), t, in.ar);
    ld_rows4_raw(blkab(P, P.BR, t, k, SZ_B), t, in.br);
    in.d = gm(P.d)[i4(P, t, k, t.L & 3)];
    in.bv = WITH_B ? ld13_raw(blkab(P, P.b, t, k, SZ_V13), t) : 0.0;
}
// v = -K x - d in lanes a < 4
template <bool WITH_B>
__device__ __forceinline__ double feedback(const Lane& t, const FwdIn<WITH_B>& in, const double x) {
    double v = t.L < 4 ? -in.d : 0.0;
    double acc = 0.0;
    double kr[13];
    mask_cols4(t, in.kr, kr);
    dotbc<13, 0>(acc, kr, x);
    v -= acc;
    settle(v);
    return v;
}
// x+ = A x + B v (+ b), all distributed; vr[4] replicated
template <bool WITH_B>
__device__ __forceinline__ double propagate(const Lane& t, const FwdIn<WITH_B>& in, const double x, const double (&vr)[4]) {
    double xn = t.L < 3 ? x : 0.0;
    if (WITH_B) xn += mask13(t, in.bv);
    double ar[10], br[4];
    mask_ar(t, in.ar, ar);
    mask_rows4(t, in.br, br);
    dotbc<10, 3>(xn, ar, x);
    SFOR(a, 0, 4, { xn += br[a] * vr[a]; });
    return xn;
}

// forward sweep of a homogeneous (delta) solve over [0, head): writes the input step to `out`.
// Three rotating stage buffers: the loads of stage k+2 are issued before the arithmetic of
// stage k (these sweeps run with ~1 wave per SIMD, so memory-level parallelism has to come from
// the wave itself).
__device__ __forceinline__ void sweep_forward_delta(const Params& P, const Lane& t, const int head, gdouble* out) {
    double x = 0.0;
    FwdIn<false> b0, b1, b2;
    auto body = [&](const FwdIn<false>& cur, int k) {
        const double dv = feedback<false>(t, cur, x);
        if (t.L < 4) out[i4(P, t, k, t.L)] = dv;
        double vr[4];
        SFOR(a, 0, 4, { vr[a] = bc<a>(dv); });
        x = propagate<false>(t, cur, x, vr);
    };
    load_fwd<false>(P, t, 0, b0);
    load_fwd<false>(P, t, imin(1, head - 1), b1);
    int k = 0;
    while (k < head) {
        load_fwd<false>(P, t, imin(k + 2, head - 1), b2);
        body(b0, k);
        if (++k >= head) break;
        load_fwd<false>(P, t, imin(k + 2, head - 1), b0);
        body(b1, k);
        if (++k >= head) break;
        load_fwd<false>(P, t, imin(k + 2, head - 1), b1);
        body(b2, k);
        ++k;
    }
}

// Backward sweep re-using the factorisation for the right-hand side P.g (input rows only);
// overwrites d.  The costate row p' is carried REPLICATED in every lane (13 registers), so only
// the row forms AR / BR / KR are needed:  rho = g + B'p ;  p' <- p'A - rho'K.
struct ResIn {
    double ar[10], br[4], kr[13], g, sv[4];
};
__device__ __forceinline__ void load_res(const Params& P, const Lane& t, const int k, ResIn& in) {
    ld_ar_raw(blkab(P, P.AR, t, k, SZ_A), t, in.ar);
    ld_rows4_raw(blkab(P, P.BR, t, k, SZ_B), t, in.br);
    ld_cols4_raw(blk(P.KR, t, P.N, k, SZ_K), t, in.kr);
    const int a = t.L & 3;
    in.g = gm(P.g)[i4(P, t, k, a)];
    const gdouble* sv = blk(P.Sinv, t, P.N, k, SZ_S) + t.q * 10;
    SFOR(c, 0, 4, {
        const int lo = a < c ? a : c, hi = a < c ? c : a;
        in.sv[c] = sv[lo * 4 - (lo * (lo - 1)) / 2 + (hi - lo)];
    });
}
__device__ __forceinline__ void sweep_resolve(const Params& P, const Lane& t, const int head) {
    double p[13];
    SFOR(j, 0, 13, { p[j] = 0.0; });
    const int a = t.L & 3;
    auto body = [&](const ResIn& cur, int k) {
        const double(&ar)[10] = cur.ar;
        const double(&br)[4] = cur.br;
        const double(&kr)[13] = cur.kr;
        double glane = t.L < 4 ? cur.g : 0.0;
        // rho[a] = g[a] + sum_l p[l] B[l][a]   (replicated)
        double rr[4], nrr[4];
        SFOR(c, 0, 4, { rr[c] = bc<c>(glane); });
        dot2bc<13, 0>(rr[0], rr[1], p, br[0], br[1]);
        dot2bc<13, 0>(rr[2], rr[3], p, br[2], br[3]);
        SFOR(c, 0, 4, { nrr[c] = -rr[c]; });
        // d = Sinv rho in lanes a < 4
        double dd = 0.0;
        SFOR(c, 0, 4, { dd += cur.sv[c] * rr[c]; });
        if (t.L < 4) gm(P.d)[i4(P, t, k, a)] = dd;
        // p' <- p'A - rho'K
        double pn[13];
        SFOR(j, 0, 3, { pn[j] = p[j]; });
        SFOR(j, 3, 13, { pn[j] = 0.0; });
        dot2bc<6, 0>(pn[3], pn[4], p, ar[0], ar[1]);
        dotbc<6, 0>(pn[5], p, ar[2]);
        dot2bc<10, 0>(pn[6], pn[7], p, ar[3], ar[4]);
        dot2bc<10, 0>(pn[8], pn[9], p, ar[5], ar[6]);
        dot2bc<13, 0>(pn[10], pn[11], p, ar[7], ar[8]);
        dotbc<13, 0>(pn[12], p, ar[9]);
        SFOR(j, 0, 13, { dotbc<4, 0>(pn[j], nrr, kr[j]); });
        SFOR(j, 0, 13, { p[j] = pn[j]; });
    };
    ResIn b0, b1, b2;
    load_res(P, t, head - 1, b0);
    load_res(P, t, imax(head - 2, 0), b1);
    int k = head - 1;
    while (k >= 0) {
        load_res(P, t, imax(k - 2, 0), b2);
        body(b0, k);
        if (--k < 0) break;
        load_res(P, t, imax(k - 2, 0), b0);
        body(b1, k);
        if (--k < 0) break;
        load_res(P, t, imax(k - 2, 0), b1);
        body(b2, k);
        --k;
    }
}

// =============================================================================================
// Primal-dual active-set solve (k_ipm, before the interior-point iteration)
// =============================================================================================
// In delta form around the unconstrained minimiser v0 (where the gradient of the condensed QP
// vanishes) the QP with a GUESSED active set is the homogeneous LQ problem (q = r = b = 0,
// dx_0 = 0) whose active inputs are fixed at c = bound - v0.  A fixed input a of stage k
//   * enters the dynamics as the affine term  b_eff = B[:, a] c  (carried by the affine row of the
//     augmented Riccati recursion, lane 13), and
//   * is taken out of the minimisation by a 1e30 on its diagonal entry of R^ (gain row, feed-
//     forward and its share of P <- M - G'K vanish to 1e-30 relative: no separate code path).
// One solve = this factorisation (which also keeps G = B'PA, rho = B'(p + P b_eff) and the rows of
// S of every stage) and ONE forward sweep: free inputs from the feedback law, fixed ones = c, and
// on the way the multiplier of every fixed input, grad = R c + B'pi_{k+1}, from the stage's own
// blocks -- the costate of the equality-constrained solve is pi = P dx + p, hence
// B'pi_{k+1} = G dx_k + (B'PB) du_free + rho: no backward costate sweep -- and the new class:
//   free   : lower / upper if v0 + du leaves the box,
//   lower  : stays while grad > 0,    upper : stays while grad < 0.
// A stationary classification satisfies the KKT conditions of the strictly convex QP exactly.
// Element state (instance-major 4-vectors of the compact slot): P.tl = c, P.tu = class (0 free,
// 1 lower, 2 upper) as a double, P.dva = du.
constexpr double AS_BIG = 1e30;
#ifndef CFN_AS_MAX
#define CFN_AS_MAX 12
#endif
#ifndef CFN_ROLL_DEPTH
#define CFN_ROLL_DEPTH 3   // (4: the same, 5: slower -- measured with the phases on opaque lane copies, 342 registers in k_as)
#endif
constexpr int AS_MAX_SOLVES = CFN_AS_MAX;   // observed on the bench workload: 48 % settle after 1 solve, 99 % within 4, all within 8
// (in two parts: the LOADS of a stage, issued one stage ahead, and what is computed from them -- R^ and b_eff -- right before
//  the stage uses them.  As one function the arithmetic sat behind the loads it had just issued, `s_waitcnt vmcnt(1)`: the
//  "prefetched" stage was waited for at once, and with it -- vector-memory operations retire in issue order -- the stores of
//  the stage before.)
__device__ __forceinline__ void load_stage_as(const Params& P, const Lane& t, const int k, StageIn<true>& in) {
    ld_ar_raw(blkab(P, P.AR, t, k, SZ_A), t, in.ar);
    ld_rows4_raw(blkab(P, P.BR, t, k, SZ_B), t, in.br);
    const int a = t.L & 3;
    in.bv = gm(P.tl)[i4(P, t, k, a)];   // c   (until finish_stage_as)
    in.Rh = gm(P.tu)[i4(P, t, k, a)];   // class
}
__device__ __forceinline__ void finish_stage_as(const Lane& t, StageIn<true>& in) {
    const double c = in.bv, cls = in.Rh;
    const double ra = t.wu;
    in.Rh = cls != 0.0 ? AS_BIG * fmax(1.0, ra) : ra;   // read in lanes a < 4 only (the 1e30 is relative to R)
    in.g = 0.0;
    in.qv = 0.0;
    const double cm = t.L < 4 ? c : 0.0;
    double bv = 0.0;
    SFOR(aa, 0, 4, { bv += in.br[aa] * bc<aa>(cm); });   // b_eff[i] = sum_a B[i][a] c_a in lane i
    in.bv = bv;
}
// Backward sweep over stages kstart .. 0.  kstart = head - 1: from the terminal cost / the checkpoint
// of the unconstrained tail; kstart < head - 1 (later solves: no input of a stage > kstart changed
// its class, so P_{kstart+1} and everything stored for the stages behind it are still valid): from
// the cost-to-go this sweep saved at stage kstart + 1 during an earlier solve.
constexpr int AS_PSAVE = 32;   // saved cost-to-go matrices per compact row (P.cPs)
// The cost-to-go is kept at every stage for short heads and at every 4th stage for heads of 24 stages and more
// (pg_shift = 0 / 2; it is 182 of the ~470 doubles a factor stage moves): a later factorisation restarts at the first
// kept stage behind the last change, i.e. repeats up to three stages (identical arithmetic, identical results).
// Measured with one granularity for all heads (k_as, every stage / 2nd / 4th): 65 536 instances 0.686 / 0.688 / 0.690 ms,
// 4096 instances 0.281 / 0.291 / 0.299 ms (the repeated stages lengthen the hardest wave's chain of short heads), kicks x 2
// 7.54 / 7.22 / 7.09 ms (long heads: the bytes count) -- hence by head.
__device__ __forceinline__ int as_pg_shift(int head) { return head >= 24 ? 2 : 0; }
__device__ __forceinline__ int as_restart_mono(int jm, int head, int sh) {   // jm = last stage whose class changed
    const int jr = (((jm >> sh) + 1) << sh) - 1;
    return (jr + 1 < head && ((jr + 1) >> sh) < AS_PSAVE) ? jr : head - 1;
}
template <bool QT = false>
__device__ __forceinline__ bool sweep_factor_as(const Params& P, const Lane& t, const int head, const int chk,
                                                const int kstart, double* wt, double* sb, const double* qtab = nullptr) {
    const int sh = as_pg_shift(head);
    double Pa[13];
    if (kstart + 1 < head) {
        const gdouble* ps = blk(P.cPs, t, AS_PSAVE, (kstart + 1) >> sh, SZ_PA) + t.q * 14 + imin(t.L, 13);
        SFOR(j, 0, 13, { Pa[j] = ps[j * 56]; });
    } else if (chk < 0) {
        SFOR(j, 0, 13, { Pa[j] = (t.L == j) ? P.WN[ext_of(j)] : 0.0; });
    } else {
        const gdouble* pc = gm(P.Pchk) + ((size_t)t.wave * N_CHK + chk) * SZ_P;
        SFOR(j, 0, 13, {
            const double v = pc[(j * 4 + t.q) * 13 + imin(t.L, 12)];
            Pa[j] = t.L < 13 ? v : 0.0;
        });
    }
    bool ok = true;
    double wq = 0.0;
    SFOR(j, 0, 13, { if (t.L == j) wq = P.W[ext_of(j)]; });
    const double is13 = t.L == 13 ? 1.0 : 0.0;
    auto keep = [&](int k) {
        if (k > 0 && (k & ((1 << sh) - 1)) == 0 && (k >> sh) < AS_PSAVE && t.L < 14) {
            gdouble* ps = blk(P.cPs, t, AS_PSAVE, k >> sh, SZ_PA) + t.q * 14 + t.L;
            SFOR(j, 0, 13, { ps[j * 56] = Pa[j]; });
        }
    };
    // three rotating stage buffers: the loads of stage k - 2 are issued before the arithmetic of stage k
    StageIn<true> b0, b1, b2;
    load_stage_as(P, t, kstart, b0);
    load_stage_as(P, t, imax(kstart - 1, 0), b1);
    int k = kstart;
    while (k >= 0) {
        load_stage_as(P, t, imax(k - 2, 0), b2);
        finish_stage_as(t, b0);
        ok = factor_stage<true, true, false, QT>(P, t, k, Pa, b0, wq, is13, wt, sb, true, qtab) && ok;
        keep(k);
        if (--k < 0) break;
        load_stage_as(P, t, imax(k - 2, 0), b0);
        finish_stage_as(t, b1);
        ok = factor_stage<true, true, false, QT>(P, t, k, Pa, b1, wq, is13, wt, sb, true, qtab) && ok;
        keep(k);
        if (--k < 0) break;
        load_stage_as(P, t, imax(k - 2, 0), b1);
        finish_stage_as(t, b2);
        ok = factor_stage<true, true, false, QT>(P, t, k, Pa, b2, wq, is13, wt, sb, true, qtab) && ok;
        keep(k);
        --k;
    }
    return ok;
}
// forward sweep: du -> P.dva; multipliers of the fixed inputs and re-classification of every input
// on the way (stage-local: grad = R c + B'pi_{k+1} with B'pi_{k+1} = G dx_k + (B'PB) du_free + rho,
// pi = P dx + p being the costate of the equality-constrained solve).  Returns the last stage of
// the row in which an input changed its class (-1: none).
// ZDX: dx_{k+1} of the solve goes to P.czdx[slot] (read by the commit kernel when the roll-out is not done in-wave)
template <bool SBOX = false, bool ZDX = false>
__device__ __forceinline__ int sweep_forward_as(const Params& P, const Lane& t, const int head) {
    // kg: lanes 0..3 hold K[a][0..12], lanes 4..7 hold G[a][0..12] -- ONE chain of 13 broadcast FMAs
    // forms the feedback (lanes 0..3) and G dx (lanes 4..7) together
    struct In { double kg[13], ar[10], br[4], d, sr[4], rho, c, cls, v0, uk, lo, hi; };
    const int a = t.L & 3;
    const bool lo4 = t.L < 4;
    auto load = [&](int k, In& in) {
        const gdouble* kb = blk(P.KR, t, P.N, k, SZ_K);
        const gdouble* gb = blk(P.cGR, t, P.N, k, SZ_K);
        const gdouble* src = (lo4 ? kb : gb) + t.q * 4 + a;
        SFOR(l, 0, 13, { in.kg[l] = src[l * 16]; });
        ld_ar_raw(blkab(P, P.AR, t, k, SZ_A), t, in.ar);      // (masked in body())
        ld_rows4_raw(blkab(P, P.BR, t, k, SZ_B), t, in.br);
        const gdouble* sr = blk(P.cS, t, P.N, k, SZ_S4) + t.q * 4 + a;
        SFOR(c, 0, 4, { in.sr[c] = sr[c * 16]; });
        const size_t idx = i4(P, t, k, a);
        in.d = gm(P.d)[idx];
        in.rho = gm(P.crho)[idx];
        in.c = gm(P.tl)[idx]; in.cls = gm(P.tu)[idx]; in.v0 = gm(P.v)[idx]; in.uk = gm(P.uit)[idx];
        box_at<SBOX>(P, idx, in.lo, in.hi);
    };
    double x = 0.0;
    int jm = -1;
    auto body = [&](const In& cur, int k) {
        double acc = 0.0;
        dotbc<13, 0>(acc, cur.kg, x);        // lanes 0..3: K[a] dx, lanes 4..7: G[a] dx
        settle(acc);
        double dv = lo4 ? -cur.d - acc : 0.0;
        dv = (lo4 && cur.cls != 0.0) ? cur.c : dv;
        if (lo4) gm(P.dva)[i4(P, t, k, t.L)] = dv;
        double gd = shift4(acc);             // lane a <- lane a + 4
        double vr[4], fr[4];
        SFOR(c, 0, 4, { vr[c] = bc<c>(dv); });
        const double dfree = cur.cls == 0.0 ? dv : 0.0;
        SFOR(c, 0, 4, { fr[c] = bc<c>(dfree); });
        SFOR(c, 0, 4, { gd += cur.sr[c] * fr[c]; });     // + (B'PB)[a][free] du_free
        if (lo4) {
            const double grad = t.wu * cur.c + gd + cur.rho;   // multiplier of a fixed input
            const double lb = cur.lo - cur.uk, ub = cur.hi - cur.uk;
            const double vn = cur.v0 + dv;
            double nc;
            if (cur.cls == 0.0) nc = vn < lb ? 1.0 : (vn > ub ? 2.0 : 0.0);
            else if (cur.cls == 1.0) nc = grad > 0.0 ? 1.0 : 0.0;
            else nc = grad < 0.0 ? 2.0 : 0.0;
            if (SBOX && !(cur.lo < cur.hi)) nc = 1.0;   // lb = ub: an equality, fixed whatever its multiplier's sign
            jm = nc != cur.cls ? k : jm;
            const size_t idx = i4(P, t, k, a);
            gm(P.tu)[idx] = nc;
            gm(P.tl)[idx] = nc == 1.0 ? lb - cur.v0 : (nc == 2.0 ? ub - cur.v0 : 0.0);
        }
        double xn = t.L < 3 ? x : 0.0;
        double ar[10], br[4];
        mask_ar(t, cur.ar, ar);
        mask_rows4(t, cur.br, br);
        dotbc<10, 3>(xn, ar, x);
        SFOR(c, 0, 4, { xn += br[c] * vr[c]; });
        x = xn;
        if (ZDX && t.L < 13) gm(P.czdx)[((size_t)t.inst * (P.N + 1) + k + 1) * 13 + t.L] = x;
    };
    // (two buffers: a third measured slower, 0.59 -> 0.605 ms in k_as)
    In b0, b1;
    load(0, b0);
    int k = 0;
    while (k < head) {
        load(imin(k + 1, head - 1), b1);
        body(b0, k);
        if (++k >= head) break;
        load(imin(k + 1, head - 1), b0);
        body(b1, k);
        ++k;
    }
    return (int)row_max((double)jm);
}

// =============================================================================================
// start solve: backward factorisation, forward sweep
// =============================================================================================
#ifndef CFN_FACTOR_DEEP
#define CFN_FACTOR_DEEP 1
#endif
// (Round 4, measured: the same sweep at THREE waves per SIMD -- a lean stage of 168 registers: affine row through the LDS tile,
//  gain by triangular substitutions, M accumulated into the new cost-to-go, scalar base addresses; 5 spilled registers --
//  passes the parity tests and runs SLOWER: 1.60 - 1.66 against 1.56 - 1.58 ms at 65 536 instances, 0.152 against 0.126 ms at
//  4096.  Occupancy is not what this kernel lacks; three stage buffers are worth 2.5 %.  profiles/r04_factor_variants.md)
KALIGN __global__ __launch_bounds__(64, 2) void k_factor(Params P) {
    __shared__ __attribute__((aligned(16))) double wtile[4][WT_TILE];
    __shared__ double btile[4][64];
    __shared__ __attribute__((aligned(16))) double qtab[16 * QT_ROW];
    const Lane t = lane_id(P);
    qtab_fill(P, qtab);
    __syncthreads();
    bool ok = sweep_factor<true, true, CFN_FACTOR_DEEP != 0>(P, t, P.N, -1, wtile[t.row], btile[t.row], 0, nullptr, false, qtab);
    ok = row_min(ok ? 1.0 : 0.0) > 0.0;
    if (t.L == 0 && t.valid) gm(P.status)[t.inst] = ok ? 0 : 4;
}

#ifdef CFN_DEV
// Stage-chunked hand-over experiment (cfnmpc_debug_chunked_pair): the stages [fk_lo, fk_hi) of
// the start solve's backward sweep, cost-to-go parked in P.Ppark between the launches, so that k_linearise can produce
// the same stages right before (its output then being read from the L2 / MALL instead of HBM -- or not: measured).
__global__ __launch_bounds__(64, 2) void k_factor_chunk(Params P) {
    __shared__ __attribute__((aligned(16))) double wtile[4][WT_TILE];
    __shared__ double btile[4][64];
    const Lane t = lane_id(P);
    const bool first = P.fk_hi >= P.N;
    gdouble* park = gm(P.Ppark) + (size_t)blockIdx.x * 13 * 64;
    bool ok = sweep_factor<true>(P, t, P.fk_hi, -1, wtile[t.row], btile[t.row], P.fk_lo, park, !first);
    ok = row_min(ok ? 1.0 : 0.0) > 0.0;
    if (t.L == 0 && t.valid) {
        if (first) gm(P.status)[t.inst] = ok ? 0 : 4;
        else if (!ok) gm(P.status)[t.inst] = 4;
    }
}
#endif

// a row whose QP failed keeps its iterate: old -> new buffers (`keep` is row-uniform; rare)
__device__ __forceinline__ void keep_row(const Params& P, const Lane& t, const bool keep) {
    if (!keep) return;
    const int N = P.N;
    const int lx = t.q * 13 + imin(t.L, 12);
    for (int k0 = 0; k0 <= N; k0 += 4) {
        double xo[4];
        SFOR(j, 0, 4, { xo[j] = blk(P.xit, t, N + 1, imin(k0 + j, N), SZ_V13)[lx]; });
        SFOR(j, 0, 4, { if (t.L < 13 && k0 + j <= N) blk(P.xitn, t, N + 1, k0 + j, SZ_V13)[lx] = xo[j]; });
    }
    for (int e0 = t.L; e0 < N * 4; e0 += 64) {   // element e = 4 k + a of the row's inputs
        double uo[4];
        SFOR(j, 0, 4, { const int e = imin(e0 + 16 * j, N * 4 - 1); uo[j] = gm(P.uit)[i4(P, t, e >> 2, e & 3)]; });
        SFOR(j, 0, 4, { const int e = e0 + 16 * j; if (e < N * 4) gm(P.uitn)[i4(P, t, e >> 2, e & 3)] = uo[j]; });
    }
}

// (compaction bins: head class x difficulty -- N_BIN, BIN_STRIDE, diff_bin in cfnmpc_ws.hpp)
// class index of a head (0: full horizon, 1..N_CHK: checkpoint stages from large to small)
__device__ __forceinline__ int head_cls(const Params& P, int h) {
    int r = 0;
    SFOR(cc, 0, N_CHK, { if (h == chk_stage(N_CHK - 1 - cc) && h < P.N) r = 1 + cc; });
    return r;
}
// head class of an instance: the interior-point sweeps must cover stages [0, want)
__device__ __forceinline__ int head_class(const Params& P, int want) {
    if (want <= 0) return 0;
    int head = P.N;
    if (P.active_horizon) {
        SFOR(c, 0, N_CHK, {
            constexpr int cs = chk_stage(N_CHK - 1 - c);
            if (want <= cs && cs < P.N) head = cs;
        });
    }
    return head;
}

// stages the constrained solve of a row must cover: its tight stages + ah_extra behind the last one -- except that the safety
// margin alone never pushes a row across the 16-stage class: a row whose tight stages END within 16 stages (last tight stage
// 12..15: 17..20 stages wanted) gets head 16, not 24.  Heads of at most 16 stages are what the head-condensed dense solves take
// (k_as_dense: one row per wavefront, ~50 us; the same row in the Riccati form of k_as_solves: 130 - 240 us, and a small fleet's
// step waits for it -- one step in three at 8192 instances; round 6: +3.3 % there, +2 % at 16 384, +1 % at
// 4096).  The tail verification (k_ascommit -> k_as_retry, or in-wave in k_as) covers the rare row for which that was too
// short: exactness does not depend on the margin.  One rule for every kernel structure, so that they keep choosing the same heads.
__device__ __forceinline__ int head_want(const Params& P, int last_tight) {
    const int want = last_tight + 1 + P.ah_extra;
    if (P.N > 16 && last_tight + 1 <= 16 && want > 16) return 16;
    return want;
}

// ---------------------------------------------------------------------------------------------
// Start solve, forward sweep -- lane-per-instance, matrix-free.
// The forward sweep only needs the PRODUCT  dx+ = A dx + B du + b, never A and B themselves, and
// that product is the directional derivative of the RK4 map at (x_k, u_k) along (dx, du): one
// forward-mode pass through the four RK stages (~10^3 flops) instead of reading the 149 stored
// entries of (A, B) per stage.  Per instance and stage the sweep then reads K (52), d (4),
// x_k (13), u_k (4) (b_k = Phi - x_{k+1} falls out of the same RK4 pass) and writes the input step
// and the candidate new state -- less than half the
// bytes of the row-distributed sweep -- and with one instance per lane there are no cross-lane
// reductions at all.  (The interior-point kernel keeps the stored A, B: its sweeps run many times
// per QP on a compact copy.)
// The CANDIDATE ITERATE x_k + dx_k, u_k + du_k goes straight into the new iterate buffers (P.xitn,
// P.uitn): for the ~92 % of the instances whose unconstrained minimiser is feasible that IS the RTI
// step (the host swaps old and new after the step); the QP kernels overwrite the others.
// ---------------------------------------------------------------------------------------------
// COND (cfnmpc_opts.cond_N2, partial condensing): the gains of a block's stages are rows of the
// CONDENSED feedback law, which acts on the state step at the START of the block (dxb); rolling the
// interior states through the stage dynamics is the `expand` step of partial condensing.
// FUSED_PT (round 4): nominal slope and directional derivative of an RK point from ONE evaluation with shared
// sub-expressions (cfnmpc_model.hpp: lf_point, 151 FP64 instructions per point instead of f_expl + jac_point + jvp's ~225).
// Where the sweep is a latency chain of N stages the shorter stream is the shorter chain (8192 - 24 576 instances: -6 %); since
// the home 4-vectors are wave-blocked it is also the faster form where the sweep streams at the HBM rate (32 768 instances
// 0.281 -> 0.254 ms, 65 536: equal), so every matrix-free sweep uses it.  (Rounds 2 - 3 kept a second form with the model's
// rotor terms as four FP64 divisions for the large fleets; removed.)
// SPLIT (small fleets, cfnmpc_opts.forward_split): the sweep in two launches -- 1: stages [0, H) with the
// classification over that window (violations, tight stages, head class, compaction ranks), hand-over of dx_H and the
// classification state in P.fs_dx / P.fs_st; 2: stages [H, N) for every instance, run BESIDE the constrained rows' QP kernels
// (whose heads of at most H stages read nothing behind H).  An instance that is feasible over [0, H) and violates a bound
// behind H is a LATE row: part 2 appends it to the compacted list behind its end (P.nipm[NI_LATE] counts them) with the head its
// tight stages ask for and the flag k_as_retry picks up (P.done = 2).
template <bool COND, bool FUSED_PT = false, int SPLIT = 0>
__device__ __forceinline__ void forward_body(const Params& P, double* xs, double* cs, int* sflag) {
    // 13-vectors travel through LDS tiles [instance][13] so that every global access of the wave
    // is a contiguous run (as in k_linearise); K, d, u, v are 32-byte runs per lane already.
    const int N = P.N;
    const int tid = threadIdx.x;
    const int raw = blockIdx.x * 64 + tid;
    const bool valid = raw < P.B;
    const int inst = valid ? raw : P.NW * 4 + (tid & 3);   // idle lanes: spare block
    const size_t w = (size_t)(inst >> 2);
    const int q = inst & 3;
    const int w0 = blockIdx.x * 16;
    const double h = P.dt;
    const double margin = P.ah_margin * (P.u_max - P.u_min);
    const gdouble* kp = gm(P.KR) + w * N * SZ_K + q * 4;
    // this lane's 4-vectors: element a of stage k at i4b + k * i4s + a (Params.v4b: wave-blocked or instance-major)
    const size_t i4b = P.v4b ? ((size_t)w * N * 4 + q) * 4 : (size_t)inst * N * 4;
    const size_t i4s = P.v4b ? 16 : 4;
    // element e = 13 * (local instance) + i of a 13-vector field with `stages` stages per block
    // (wave-uniform 64-bit base + 32-bit byte offset per lane: saddr form of the global access)
    auto el13 = [&](const double* f, int e, int stages, int k) -> gdouble* {
        const int bk = (int)(__umul24((unsigned)e, div_magic(52)) >> 16), off = e - bk * 52;   // e / 52 (24-bit multiplies, as in k_linearise)
        const unsigned bo = (__umul24((unsigned)imin(bk, P.NW - w0), (unsigned)(stages * SZ_V13)) + (unsigned)off) * 8u;
        const char* base = (const char*)(gm(f) + ((size_t)w0 * stages + k) * SZ_V13);
        return (gdouble*)(base + bo);
    };
    auto issue13 = [&](const double* f, int stages, int k, int tl, double (&r)[13]) {
        SFOR(j, 0, 13, { r[j] = *el13(f, tl + 64 * j, stages, k); });
    };
    auto land13 = [&](double* tile, const double (&r)[13]) {
        SFOR(j, 0, 13, { tile[tid + 64 * j] = r[j]; });
    };
    struct In { double K[4][13], d[4], u[4]; };   // K: column index in INTERNAL order
    auto load = [&](int k, In& in) {
        SFOR(l, 0, 13, { SFOR(a, 0, 4, { in.K[a][l] = kp[(size_t)k * SZ_K + l * 16 + a]; }); });
        SFOR(a, 0, 4, { in.d[a] = gm(P.d)[i4b + (size_t)k * i4s + a]; in.u[a] = gm(P.uit)[i4b + (size_t)k * i4s + a]; });
    };
    const int H = SPLIT ? P.fwd_split : 0;
    const int k_lo = SPLIT == 2 ? H : 0, k_hi = SPLIT == 1 ? H : N;
    double dx[13];   // internal order
    double viol = 0.0;
    int last_tight = -1, nviol = 0;
    bool sawnan = false;
    bool early_infeasible = false, early_bad = false;   // SPLIT == 2: what part one decided
    if constexpr (SPLIT == 2) {
        // hand-over of part one: dx_H and the classification state, [group][13 | 4][64 lanes]
        const gdouble* hx = gm(P.fs_dx) + (size_t)blockIdx.x * 13 * 64 + tid;
        SFOR(i, 0, 13, { dx[i] = hx[i * 64]; });
        const gint* hs = gm(P.fs_st) + (size_t)blockIdx.x * 4 * 64 + tid;
        early_infeasible = hs[0] != 0; early_bad = hs[64] != 0;
        double r1[13];
        issue13(P.xit, N + 1, k_lo, tid, r1);
        land13(xs, r1);
        __syncthreads();
    } else {
        double r0[13], r1[13];
        issue13(P.x0, 1, 0, tid, r0);
        issue13(P.xit, N + 1, 0, tid, r1);
        land13(cs, r0);
        land13(xs, r1);
        __syncthreads();
        SFOR(i, 0, 13, { dx[i] = cs[tid * 13 + i] - xs[tid * 13 + i]; });
        __syncthreads();
    }
    In cur, nxt;
    load(k_lo, cur);
    // (waited for HERE: the compiler sizes the wait at the loop's top for the worse of its two entries, and from this side nothing
    //  younger than the loads is in flight -- it would be vmcnt(0), i.e. every stage would also wait for the 13 state stores the
    //  stage before issued behind its loads)
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    double dxb[13];            // COND: state step at the start of the current block
    int knext = 0, jblk = 0;   // COND: first stage and index of the next block
    for (int k = k_lo; k < k_hi; k++) {
        int tl = tid;   // opaque per-stage copy (keeps the transfer offsets out of loop-invariant registers)
        asm volatile("" : "+v"(tl));
        if constexpr (COND) if (k == knext) {
            SFOR(i, 0, 13, { dxb[i] = dx[i]; });
            knext += cond_len(P, jblk++);
        }
        double xi[13];
        SFOR(i, 0, 13, { xi[i] = xs[tid * 13 + i]; });
        // candidate state of stage k
        SFOR(i, 0, 13, { cs[tid * 13 + i] = xi[i] + dx[i]; });
        // du = -K dx - d, bounds
        double du[4];
        SFOR(a, 0, 4, {
            double acc = 0.0;
            SFOR(l, 0, 13, { acc = __builtin_fma(cur.K[a][l], COND ? dxb[l] : dx[l], acc); });
            du[a] = -cur.d[a] - acc;
            const double lb = P.u_min - cur.u[a], ub = P.u_max - cur.u[a];
            viol = fmax(viol, fmax(lb - du[a], du[a] - ub));
            nviol += (du[a] < lb || du[a] > ub) ? 1 : 0;
            if (du[a] < lb + margin || du[a] > ub - margin) last_tight = k;
            sawnan = sawnan || !(du[a] == du[a]);
            gm(P.v)[i4b + (size_t)k * i4s + a] = du[a];
            gm(P.uitn)[i4b + (size_t)k * i4s + a] = cur.u[a] + du[a];
        });
        // next stage's inputs (issued here: the gain of stage k is dead, its registers are free)
        const double uc[4] = {cur.u[0], cur.u[1], cur.u[2], cur.u[3]};
        load(imin(k + 1, N - 1), nxt);
        double xr[13];
        issue13(P.xit, N + 1, k + 1, tl, xr);
        // directional derivative of the RK4 step along (dx, du); model vectors in EXTERNAL order
        double x[13], s[13], xt[13], st[13], kk[13], dk[13], acc[13], ks[13];   // ks = k1 + 2 k2 + 2 k3 + k4 (nominal)
        SFOR(e, 0, 13, { x[e] = xi[int_of(e)]; s[e] = dx[int_of(e)]; });
        double jud[4];
        {
            const double p0 = uc[0] * du[0], p1 = uc[1] * du[1], p2 = uc[2] * du[2], p3 = uc[3] * du[3];
            jud[0] = 2.0 * KT * (p0 + p1 + p2 + p3);
            jud[1] = 2.0 * KA * (p0 + p1 - p2 - p3);
            jud[2] = 2.0 * KB * (p0 - p1 - p2 + p3);
            jud[3] = 2.0 * KC * (p0 - p1 + p2 - p3);
        }
        double dxp[13];   // dx_{k+1} + x_{k+1}
        if constexpr (FUSED_PT) {
            double rot[4];
            {
                const double s1 = uc[0] * uc[0], s2 = uc[1] * uc[1], s3 = uc[2] * uc[2], s4 = uc[3] * uc[3];
                rot[0] = KT * (s1 + s2 + s3 + s4);
                rot[1] = KA * (s1 + s2 - s3 - s4);
                rot[2] = KB * (s1 - s2 - s3 + s4);
                rot[3] = KC * (s1 - s2 + s3 - s4);
            }
            double xq[10], sq[10];
            SFOR(e, 0, 10, { xq[e] = x[e + 3]; sq[e] = s[e + 3]; });
            lf_point(xq, sq, rot, jud, kk, dk);
            SFOR(e, 0, 13, { acc[e] = dk[e]; ks[e] = kk[e]; });
            SFOR(e, 0, 10, { xq[e] = x[e + 3] + 0.5 * h * kk[e + 3]; sq[e] = s[e + 3] + 0.5 * h * dk[e + 3]; });
            lf_point(xq, sq, rot, jud, kk, dk);
            SFOR(e, 0, 13, { acc[e] += 2.0 * dk[e]; ks[e] += 2.0 * kk[e]; });
            SFOR(e, 0, 10, { xq[e] = x[e + 3] + 0.5 * h * kk[e + 3]; sq[e] = s[e + 3] + 0.5 * h * dk[e + 3]; });
            lf_point(xq, sq, rot, jud, kk, dk);
            SFOR(e, 0, 13, { acc[e] += 2.0 * dk[e]; ks[e] += 2.0 * kk[e]; });
            SFOR(e, 0, 10, { xq[e] = x[e + 3] + h * kk[e + 3]; sq[e] = s[e + 3] + h * dk[e + 3]; });
            lf_point(xq, sq, rot, jud, kk, dk);
        } else {
        JacPoint J;
        // stage 1
        f_expl(x, uc, kk);
        jac_point(x, J);
        jvp<true, true>(J, s, dk);
        SFOR(i, 0, 4, { dk[9 + i] += jud[i]; });
        SFOR(e, 0, 13, { acc[e] = dk[e]; ks[e] = kk[e]; xt[e] = x[e] + 0.5 * h * kk[e]; st[e] = s[e] + 0.5 * h * dk[e]; });
        // stage 2
        f_expl(xt, uc, kk);
        jac_point(xt, J);
        jvp<true, true>(J, st, dk);
        SFOR(i, 0, 4, { dk[9 + i] += jud[i]; });
        SFOR(e, 0, 13, { acc[e] += 2.0 * dk[e]; ks[e] = ks[e] + 2 * kk[e]; xt[e] = x[e] + 0.5 * h * kk[e]; st[e] = s[e] + 0.5 * h * dk[e]; });
        // stage 3
        f_expl(xt, uc, kk);
        jac_point(xt, J);
        jvp<true, true>(J, st, dk);
        SFOR(i, 0, 4, { dk[9 + i] += jud[i]; });
        SFOR(e, 0, 13, { acc[e] += 2.0 * dk[e]; ks[e] = ks[e] + 2 * kk[e]; xt[e] = x[e] + h * kk[e]; st[e] = s[e] + h * dk[e]; });
        // stage 4 (the nominal slope too: b_k = Phi(x_k, u_k) - x_{k+1} is formed here, as k_linearise
        // forms it, instead of being read back)
        f_expl(xt, uc, kk);
        jac_point(xt, J);
        jvp<true, true>(J, st, dk);
        SFOR(i, 0, 4, { dk[9 + i] += jud[i]; });
        }
        SFOR(i, 0, 13, {
            constexpr int e = ext_of(i);
            const double phi = x[e] + (h / 6.0) * (ks[e] + kk[e]);
            dxp[i] = s[e] + (h / 6.0) * (acc[e] + dk[e]) + phi;
        });
        // candidate tile out, next stage's tiles in
        __syncthreads();
        {
            double cv[13];
            SFOR(j, 0, 13, { cv[j] = cs[tl + 64 * j]; });
            SFOR(j, 0, 13, { *el13(P.xitn, tl + 64 * j, N + 1, k) = cv[j]; });
        }
        land13(xs, xr);
        __syncthreads();
        SFOR(i, 0, 13, { dx[i] = dxp[i] - xs[tid * 13 + i]; });
        cur = nxt;
    }
    if (sawnan) viol = nan("");
    if constexpr (SPLIT == 1) {
        // ---- part one: classification over [0, H), hand-over; no terminal candidate, no restoration (part two restores the
        //      whole horizon of a failed instance -- nobody reads its new iterate in between: it is in no list)
        const bool okf = valid && gm(P.status)[imin(raw, P.B - 1)] == 0;
        const bool bad = valid && (!okf || !(viol == viol));
        const bool infeasible = valid && !bad && (viol > 0.0);
        gdouble* hx = gm(P.fs_dx) + (size_t)blockIdx.x * 13 * 64 + tid;
        SFOR(i, 0, 13, { hx[i * 64] = dx[i]; });
        gint* hs = gm(P.fs_st) + (size_t)blockIdx.x * 4 * 64 + tid;
        hs[0] = infeasible ? 1 : 0; hs[64] = bad ? 1 : 0;
        if (valid) {
            gm(P.viol)[inst] = infeasible ? viol : 0.0;
            gm(P.status)[inst] = bad ? 4 : 0;
            gm(P.iters)[inst] = 0;
            gm(P.res)[inst] = bad ? nan("") : 0.0;
            gm(P.head)[inst] = infeasible ? head_class(P, head_want(P, last_tight)) : 0;
            if (P.as_warm && !infeasible) gm(P.wvalid)[inst] = 0;
        }
        const int hc = infeasible ? head_cls(P, head_class(P, head_want(P, last_tight))) * N_DIFF + diff_bin(nviol, P.as_passes == 0 && P.active_set) : -1;
        const unsigned long long below = (1ull << tid) - 1ull;
        SFOR(c, 0, N_BIN, {
            const unsigned long long m = __ballot(hc == c);
            if (hc == c) gm(P.rank)[inst] = (c << 8) | __popcll(m & below);
            if (tid == c) gm(P.blkcnt)[blockIdx.x * BIN_STRIDE + c] = __popcll(m);
        });
        return;
    }
    // candidate of the terminal stage (xs holds x_N)
    SFOR(i, 0, 13, { cs[tid * 13 + i] = xs[tid * 13 + i] + dx[i]; });
    bool bad, infeasible;
    if constexpr (SPLIT == 2) {
        // ---- part two: an instance that failed here only (non-finite candidate behind H; it was feasible and healthy before)
        //      fails as a whole; one that was feasible over [0, H) and leaves the box behind H is a LATE row
        const bool bad2 = valid && !early_bad && !early_infeasible && !(viol == viol);
        bad = early_bad || bad2;
        infeasible = false;   // (nothing of part one's classification is rewritten)
        const bool late = valid && !bad && !early_infeasible && (viol > 0.0);
        if (bad2) { gm(P.status)[inst] = 4; gm(P.res)[inst] = nan(""); }
        if (late) {
            gm(P.viol)[inst] = viol;
            gm(P.head)[inst] = head_class(P, head_want(P, last_tight));
            gm(P.done)[inst] = 2;                                   // k_as_retry solves it (the interior point what that leaves)
            const int pos = atomicAdd(P.nipm + NI_LATE, 1);
            gm(P.ilist)[gm(P.nipm)[0] + pos] = inst;
        }
    } else {
        const bool okf = valid && gm(P.status)[imin(raw, P.B - 1)] == 0;
        bad = valid && (!okf || !(viol == viol));
        infeasible = valid && !bad && (viol > 0.0);
    }
    sflag[tid] = bad ? 1 : 0;
    __syncthreads();
    {
        double cv[13];
        SFOR(j, 0, 13, { cv[j] = cs[tid + 64 * j]; });
        SFOR(j, 0, 13, { *el13(P.xitn, tid + 64 * j, N + 1, N) = cv[j]; });
    }
    if constexpr (SPLIT == 0) {
    if (valid) {
        gm(P.viol)[inst] = infeasible ? viol : 0.0;
        gm(P.status)[inst] = bad ? 4 : 0;
        gm(P.iters)[inst] = 0;
        gm(P.res)[inst] = bad ? nan("") : 0.0;
        gm(P.head)[inst] = infeasible ? head_class(P, head_want(P, last_tight)) : 0;
        if (P.as_warm && !infeasible) gm(P.wvalid)[inst] = 0;   // an unconstrained step ends the instance's run of constrained ones
    }
    {   // first half of the stable compaction: per-group bin counts and ranks.  Bin = head class
        // (largest first) x difficulty (number of violated inputs of the unconstrained minimiser in
        // N_DIFF classes, diff_bin -- the active-set solve needs more passes the more bounds are involved,
        // and a wave lasts as long as the slowest of its four rows)
        const int hc = infeasible ? head_cls(P, head_class(P, head_want(P, last_tight))) * N_DIFF + diff_bin(nviol, P.as_passes == 0 && P.active_set) : -1;
        const unsigned long long below = (1ull << tid) - 1ull;
        SFOR(c, 0, N_BIN, {
            const unsigned long long m = __ballot(hc == c);
            if (hc == c) gm(P.rank)[inst] = (c << 8) | __popcll(m & below);
            if (tid == c) gm(P.blkcnt)[blockIdx.x * BIN_STRIDE + c] = __popcll(m);
        });
    }
    }
    // The candidate went straight into the NEW iterate buffers (P.xitn, P.uitn; the host swaps the
    // buffers after the step), so instances whose unconstrained minimiser is feasible are done.
    // Constrained ones are overwritten there by the QP kernels (or restored if their QP fails);
    // the rare failed instance (NaN / failed factorisation) keeps its iterate: copy old -> new.
    // The wave copies cooperatively, element e belonging to instance e / 13.
    if (__any(bad)) {
        bool mine[13];
        SFOR(j, 0, 13, { mine[j] = sflag[(tid + 64 * j) / 13] != 0; });
        for (int k0 = 0; k0 <= N; k0 += 2) {
            double c[2][13];
            SFOR(jj, 0, 2, { SFOR(j, 0, 13, { c[jj][j] = *el13(P.xit, tid + 64 * j, N + 1, imin(k0 + jj, N)); }); });
            SFOR(jj, 0, 2, {
                if (k0 + jj <= N) SFOR(j, 0, 13, { if (mine[j]) *el13(P.xitn, tid + 64 * j, N + 1, k0 + jj) = c[jj][j]; });
            });
        }
    }
    if (bad) {
        for (int k = 0; k < N; k++) SFOR(a, 0, 4, { gm(P.uitn)[i4b + (size_t)k * i4s + a] = gm(P.uit)[i4b + (size_t)k * i4s + a]; });
    }
}

KALIGN __global__ __launch_bounds__(64) void k_forward(Params P) {   // matrix-free forward sweep of the start solve
    __shared__ double xs[64 * 13], cs[64 * 13];
    __shared__ int sflag[64];
    forward_body<false, true>(P, xs, cs, sflag);
}
#ifdef CFN_DEV
// the same at two waves per SIMD (<= 256 registers, a few spills): beside the fused start solve of ANOTHER sub-fleet, whose
// waves leave half a SIMD's register file each (cfnmpc_opts.sub_fleets)
__global__ __launch_bounds__(64, 2) void k_forward_half(Params P) {
    __shared__ double xs[64 * 13], cs[64 * 13];
    __shared__ int sflag[64];
    forward_body<false, true>(P, xs, cs, sflag);
}
#endif
KALIGN __global__ __launch_bounds__(64) void k_forward_p1(Params P) {   // split sweep, stages [0, H) + classification
    __shared__ double xs[64 * 13], cs[64 * 13];
    __shared__ int sflag[64];
    forward_body<false, true, 1>(P, xs, cs, sflag);
}
KALIGN __global__ __launch_bounds__(64) void k_forward_p2(Params P) {   // split sweep, stages [H, N), late rows
    __shared__ double xs[64 * 13], cs[64 * 13];
    __shared__ int sflag[64];
    forward_body<false, true, 2>(P, xs, cs, sflag);
}
__global__ __launch_bounds__(64) void k_cforward(Params P) {
    __shared__ double xs[64 * 13], cs[64 * 13];
    __shared__ int sflag[64];
    forward_body<true>(P, xs, cs, sflag);
}

// ---------------------------------------------------------------------------------------------
// Start solve, forward sweep on the STORED stage blocks -- row groups, four instances per wave.
// Same results as k_forward up to rounding (dx+ = A dx + B du + b from the stored A, B, b instead of
// the directional derivative of the RK4 map).  For SMALL fleets: k_forward runs 64 instances per wave,
// i.e. B / 64 waves of 50 sequential stages at ~4 us each -- with a few thousand instances most SIMDs
// idle and the step waits for that chain; here a stage costs one 13-term broadcast-FMA chain per
// product (~1.5 us) and the blocks it reads were written two kernels ago (L2 / MALL-resident at this
// size).  At large batches the streaming of A and B costs more than the arithmetic saves
// (DESIGN.md section 5.4), so the choice is by batch size (cfnmpc_opts.forward_sweep).
// The compaction ranks (per 64-instance group, as k_forward leaves them) come from k_rank.
// ---------------------------------------------------------------------------------------------
template <bool SBOX, int DEPTH = 3>
__device__ __forceinline__ void forward_rg_body(const Params& P) {
    const Lane t = lane_id(P);
    const int N = P.N;
    const double margin = P.ah_margin * (P.u_max - P.u_min);
    const int a = t.L & 3;
    const bool lo4 = t.L < 4;
    // DEPTH rotating stage buffers: the loads of stage k + DEPTH - 1 are issued before the arithmetic of stage k
    // (small fleets run about one wave per SIMD: memory-level parallelism has to come from the wave itself;
    // measured on fleets of 1024 / 2048 / 4096 instances, one wave per SIMD: DEPTH 3: 40 / 51 / 106 us, 4: 42 / 55 / 108,
    // 5: 47 / 60 / 112, 6: 50 / 62 / 114 -- the buffers beyond the VGPR file are parked in AGPRs, and the copy waits
    // for the load; at 4096 instances the sweep moves 370 MB in 106 us = 3.5 TB/s: no longer a latency chain)
    struct In { FwdIn<true> f; double u, xb, lo, hi; };
    auto load = [&](int k, In& in) {
        load_fwd<true>(P, t, k, in.f);
        in.u = gm(P.uit)[i4(P, t, k, a)];
        box_at<SBOX>(P, i4(P, t, k, a), in.lo, in.hi);
        in.xb = ld13_raw(blk(P.xit, t, N + 1, k + 1, SZ_V13), t);   // (stored by st13: lanes < 13 only)
    };
    double xbcur = ld13(blk(P.xit, t, N + 1, 0, SZ_V13), t);
    double x = ld13(blk(P.x0, t, 1, 0, SZ_V13), t) - xbcur;
    double viol = 0.0;
    int last_tight = -1, nviol = 0;
    bool sawnan = false;
    auto body = [&](const In& cur, int k) {
        st13(blk(P.xitn, t, N + 1, k, SZ_V13), t, xbcur + x);
        const double v = feedback<true>(t, cur.f, x);      // lanes a < 4: du = -K dx - d
        if (lo4) {
            const double lb = cur.lo - cur.u, ub = cur.hi - cur.u;
            viol = fmax(viol, fmax(lb - v, v - ub));
            nviol += (v < lb || v > ub) ? 1 : 0;
            if (v < lb + margin || v > ub - margin) last_tight = k;
            sawnan = sawnan || !(v == v);
            gm(P.v)[i4(P, t, k, t.L)] = v;
            gm(P.uitn)[i4(P, t, k, t.L)] = cur.u + v;
        }
        double vr[4];
        SFOR(c, 0, 4, { vr[c] = bc<c>(v); });
        x = propagate<true>(t, cur.f, x, vr);
        xbcur = cur.xb;
    };
    {   // DEPTH rotating stage buffers (compile-time indices: registers)
        In b[DEPTH];
        SFOR(j, 0, DEPTH - 1, { load(imin(j, N - 1), b[j]); });
        int k = 0;
        while (k < N) {
            SFOR(j, 0, DEPTH, {
                if (k < N) {
                    load(imin(k + DEPTH - 1, N - 1), b[(j + DEPTH - 1) % DEPTH]);
                    body(b[j], k);
                    ++k;
                }
            });
        }
    }
    st13(blk(P.xitn, t, N + 1, N, SZ_V13), t, xbcur + x);
    // reductions over the four input lanes of the row
    viol = row_max(lo4 ? viol : 0.0);
    nviol = (int)row_sum(lo4 ? (double)nviol : 0.0);
    last_tight = (int)row_max(lo4 ? (double)last_tight : -1.0);
    sawnan = row_max((lo4 && sawnan) ? 1.0 : 0.0) > 0.0;
    if (sawnan) viol = nan("");
    const bool okf = t.valid && gm(P.status)[imin(t.inst, P.B - 1)] == 0;
    const bool bad = t.valid && (!okf || !(viol == viol));
    const bool infeasible = t.valid && !bad && (viol > 0.0);
    if (t.L == 0 && t.valid) {
        const int head = infeasible ? head_class(P, head_want(P, last_tight)) : 0;
        gm(P.viol)[t.inst] = infeasible ? viol : 0.0;
        gm(P.status)[t.inst] = bad ? 4 : 0;
        gm(P.iters)[t.inst] = 0;
        gm(P.res)[t.inst] = bad ? nan("") : 0.0;
        gm(P.head)[t.inst] = head;
        if (P.as_warm && !infeasible) gm(P.wvalid)[t.inst] = 0;
        // compaction bin (head class x difficulty), ranked per 64-instance group by k_rank
        gm(P.rank)[t.inst] = infeasible ? head_cls(P, head) * N_DIFF + diff_bin(nviol, P.as_passes == 0 && P.active_set) : -1;
    }
    keep_row(P, t, bad);   // a failed row keeps its iterate: old -> new
}
__global__ __launch_bounds__(64) void k_forward_rg(Params P) { forward_rg_body<false>(P); }
__global__ __launch_bounds__(64) void k_forward_rg_sbox(Params P) { forward_rg_body<true>(P); }
// per 64-instance group: bin counts and ranks of the constrained instances (the second half of
// k_forward's epilogue, for the row-group forward sweep)
__global__ __launch_bounds__(64) void k_rank(Params P) {
    const int tid = threadIdx.x;
    const int raw = blockIdx.x * 64 + tid;
    const int hc = raw < P.B ? gm(P.rank)[raw] : -1;
    const unsigned long long below = (1ull << tid) - 1ull;
    SFOR(c, 0, N_BIN, {
        const unsigned long long m = __ballot(hc == c);
        if (hc == c) gm(P.rank)[raw] = (c << 8) | __popcll(m & below);
        if (tid == c) gm(P.blkcnt)[blockIdx.x * BIN_STRIDE + c] = __popcll(m);
    });
}

// Stable compaction of the instances that need the interior-point method, grouped by head
// class (largest first) so that the four rows of a wave work on similar horizons.  k_forward
// left per-group class counts and per-instance ranks; k_compact (one block) turns the counts
// into per-group bases, k_scatter places every instance.
// k_compact: one workgroup per bin; thread t owns a run of consecutive groups, the workgroup scans
// its bin's counts over the groups (exclusive prefix, in place) and leaves the bin's total in
// P.nipm[1 + bin].  k_scatter adds the totals of the preceding bins.
__global__ __launch_bounds__(256) void k_compact(Params P) {
    constexpr int NT = 256;
    __shared__ int cnt[NT];
    const int tid = threadIdx.x, j = blockIdx.x;
    const int ng = (P.B + 63) / 64;                 // groups of k_forward
    const int chunk = (ng + NT - 1) / NT;
    const int lo = tid * chunk, hi = min(lo + chunk, ng);
    int c = 0;
    for (int g = lo; g < hi; g++) c += gm(P.blkcnt)[g * BIN_STRIDE + j];
    cnt[tid] = c;
    __syncthreads();
    for (int off = 1; off < NT; off <<= 1) {        // inclusive scan over threads (Hillis-Steele)
        const int v = tid >= off ? cnt[tid - off] : 0;
        __syncthreads();
        cnt[tid] += v;
        __syncthreads();
    }
    if (tid == NT - 1) gm(P.nipm)[1 + j] = cnt[tid];
    int pos = cnt[tid] - c;
    for (int g = lo; g < hi; g++) {
        const int n = gm(P.blkcnt)[g * BIN_STRIDE + j];
        gm(P.blkcnt)[g * BIN_STRIDE + j] = pos;
        pos += n;
    }
}
__global__ __launch_bounds__(256) void k_scatter(Params P) {
    __shared__ int base[N_BIN + 1];
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int j = 0; j < N_BIN; j++) { base[j] = acc; acc += gm(P.nipm)[1 + j]; }
        base[N_BIN] = acc;
        if (blockIdx.x == 0) {
            gm(P.nipm)[0] = acc;
            // rows whose head is longer than 16 stages come first in the list (bins are ordered by head class, longest first:
            // full horizon, 32, 24 | 16, 12, 8, 4): their number -- the dense active-set kernel (cfnmpc_asdense.hip) takes the rest
            gm(P.nipm)[NI_LONG16] = P.N > 16 ? base[3 * N_DIFF] : 0;
            gm(P.nipm)[NI_LONG24] = P.N > 24 ? base[2 * N_DIFF] : 0;    // ... of more than 24 stages (full horizon, 32): behind part two of a split sweep
            gm(P.nipm)[NI_LATE] = 0;                                    // late rows of this step's split sweep (k_forward_p2 counts them)
            gm(P.nipm)[NI_LISTED] = 0;                                  // fall-back rows (the monolithic active-set kernel appends them; k_ipm_list overwrites)
        }
    }
    if (blockIdx.x == 0 && P.ascnt && threadIdx.x < 32) gm(P.ascnt)[threadIdx.x] = 0;   // work lists of the active-set passes
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.B) return;
    if (gm(P.head)[i] > 0) {
        const int r = gm(P.rank)[i];
        const int pos = base[r >> 8] + gm(P.blkcnt)[(i >> 6) * BIN_STRIDE + (r >> 8)] + (r & 255);
        gm(P.ilist)[pos] = i;
        // split forward sweep: a row whose head reaches behind the split point (classes 32 / N: less than one per step at these
        // fleet sizes) cannot be solved beside part two of the sweep -- it goes to the retry kernel, which runs behind the commit
        // kernel anyway (P.asst = 2: "not tried"; a launch that does solve it, k_as_solves over the whole list, overwrites the flag)
        if (P.fwd_split && P.N > 24 && (r >> 8) < 2 * N_DIFF) gm(P.asst)[pos] = 2;
    }
}

// =============================================================================================
// interior-point QP on the waves that need it
// =============================================================================================
struct RowIPM {  // uniform over the 16 lanes of a row
    double mu, res;
    int iters, status;
    bool act;
};

__device__ __forceinline__ double ratio(double z, double dz, double a) {
    const double tt = -z * rcp_nr(dz);
    return (dz < 0.0 && tt < a) ? tt : a;
}

// element-wise passes: lane L of a row handles elements e = L, L+16, ... of the head*4 inputs
struct Elem {
    double v, tl, tu, ll, lu, rg, lb, ub;
    double dva, dvc;   // predictor / corrector input steps (passes that need them)
};
template <int NSTEP, bool SBOX = false>   // NSTEP = 0: state only, 1: + dva, 2: + dva, dvc
__device__ __forceinline__ Elem ld_elem(const Params& P, size_t idx) {
    Elem e;
    e.v = gm(P.v)[idx]; e.tl = gm(P.tl)[idx]; e.tu = gm(P.tu)[idx]; e.ll = gm(P.ll)[idx]; e.lu = gm(P.lu)[idx]; e.rg = gm(P.rg)[idx];
    const double uk = gm(P.uit)[idx];
    double lo, hi;
    box_at<SBOX>(P, idx, lo, hi);
    e.lb = lo - uk;
    e.ub = hi - uk;
    e.dva = NSTEP >= 1 ? gm(P.dva)[idx] : 0.0;
    e.dvc = NSTEP >= 2 ? gm(P.dvc)[idx] : 0.0;
    return e;
}
// One pass over the n elements of a row, in batches of four per lane with all loads of a batch
// issued before any arithmetic (these passes run with one wave per SIMD: a plain loop would pay
// one memory round trip per element).
#ifndef CFN_ELEM_PIPE
#define CFN_ELEM_PIPE 1
#endif
template <int NSTEP, bool SBOX, class BODY>
__device__ __forceinline__ void elem_pass(const Params& P, size_t base, int n, int L, BODY&& body) {
    if constexpr (CFN_ELEM_PIPE && !SBOX) {   // (per-stage boxes load two more values per element: that variant keeps the plain batches, +36 B of scratch otherwise)
    // two half-batches in flight (the same four elements' worth of registers): the loads of the next two elements are issued before
    // the arithmetic of the current two, so that a pass over a long row (200 inputs at N = 50: 13 elements per lane) pays about
    // one memory round trip instead of one per batch of four -- the fall-back's five passes per iteration were 25 of its 194 us
    Elem cur[2], nxt[2];
    SFOR(j, 0, 2, { cur[j] = ld_elem<NSTEP, SBOX>(P, base + imin(L + 16 * j, n - 1)); });
    for (int e0 = L; e0 < n; e0 += 32) {
        SFOR(j, 0, 2, { nxt[j] = ld_elem<NSTEP, SBOX>(P, base + imin(e0 + 32 + 16 * j, n - 1)); });
        SFOR(j, 0, 2, { if (e0 + 16 * j < n) body(e0 + 16 * j, cur[j]); });
        SFOR(j, 0, 2, { cur[j] = nxt[j]; });
    }
    } else {
    for (int e0 = L; e0 < n; e0 += 64) {
        Elem d[4];
        SFOR(j, 0, 4, { d[j] = ld_elem<NSTEP, SBOX>(P, base + imin(e0 + 16 * j, n - 1)); });
        SFOR(j, 0, 4, { if (e0 + 16 * j < n) body(e0 + 16 * j, d[j]); });
    }
    }
}

// ---------------------------------------------------------------------------------------------
// Clipped start of the interior point (cfnmpc_opts.ipm_clip_viol): gradient
// of the condensed head QP at v0 + dv, g = H dv, by one forward sweep (dx_k -> P.czdx of the compact slot) and one
// backward costate sweep (pi_head = P_head dx_head; g_k = R dv_k + B'pi_{k+1}; pi_k = Q dx_k + A'pi_{k+1}, the costate
// carried replicated as in sweep_resolve).  dv is read from Q.dva, g is left in Q.g.  Rows with dv = 0 get g = 0.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void sweep_clip_gradient(const Params& P, const Params& Q, const Lane& tc, const int head, const int chk) {
    const int N = P.N;
    const int a = tc.L & 3;
    const bool lo4 = tc.L < 4;
    gdouble* zx = gm(P.czdx) + (size_t)tc.inst * (N + 1) * 13;
    {   // forward: dx_0 = 0, dx_{k+1} = A dx_k + B dv_k
        struct In { double ar[10], br[4], dv; };
        auto load = [&](int k, In& in) {
            ld_ar(blkab(Q, Q.AR, tc, k, SZ_A), tc, in.ar);
            ld_rows4(blkab(Q, Q.BR, tc, k, SZ_B), tc, in.br);
            in.dv = gm(Q.dva)[i4(Q, tc, k, a)];
        };
        double x = 0.0;
        if (tc.L < 13) zx[tc.L] = 0.0;
        auto body = [&](const In& cur, int k) {
            const double dvl = lo4 ? cur.dv : 0.0;
            double vr[4];
            SFOR(c, 0, 4, { vr[c] = bc<c>(dvl); });
            double xn = tc.L < 3 ? x : 0.0;
            dotbc<10, 3>(xn, cur.ar, x);
            SFOR(c, 0, 4, { xn += cur.br[c] * vr[c]; });
            x = xn;
            if (tc.L < 13) zx[(size_t)(k + 1) * 13 + tc.L] = x;
        };
        In b0, b1;
        load(0, b0);
        int k = 0;
        while (k < head) {
            load(imin(k + 1, head - 1), b1);
            body(b0, k);
            if (++k >= head) break;
            load(imin(k + 1, head - 1), b0);
            body(b1, k);
            ++k;
        }
    }
    // backward: costate replicated in every lane
    double p[13];
    {
        double xh[13];
        SFOR(j, 0, 13, { xh[j] = zx[(size_t)head * 13 + j]; });
        if (chk < 0) {
            SFOR(j, 0, 13, { p[j] = P.WN[ext_of(j)] * xh[j]; });
        } else {   // pi = P_head dx_head, P_head row-distributed (lane i holds row i)
            const gdouble* pc = gm(Q.Pchk) + ((size_t)tc.wave * N_CHK + chk) * SZ_P;
            double pd = 0.0;
            SFOR(j, 0, 13, { pd += pc[(j * 4 + tc.q) * 13 + imin(tc.L, 12)] * xh[j]; });
            SFOR(j, 0, 13, { p[j] = bc<j>(pd); });
        }
    }
    struct Bk { double ar[10], br[4], dv, xk[13]; };
    auto loadb = [&](int k, Bk& in) {
        ld_ar_raw(blkab(Q, Q.AR, tc, k, SZ_A), tc, in.ar);
        ld_rows4_raw(blkab(Q, Q.BR, tc, k, SZ_B), tc, in.br);
        in.dv = gm(Q.dva)[i4(Q, tc, k, a)];
        SFOR(j, 0, 13, { in.xk[j] = zx[(size_t)k * 13 + j]; });
    };
    auto bodyb = [&](const Bk& cur, int k) {
        const double(&ar)[10] = cur.ar;
        const double(&br)[4] = cur.br;
        double glane = lo4 ? tc.wu * cur.dv : 0.0;
        double rr[4];
        SFOR(c, 0, 4, { rr[c] = bc<c>(glane); });
        dot2bc<13, 0>(rr[0], rr[1], p, br[0], br[1]);
        dot2bc<13, 0>(rr[2], rr[3], p, br[2], br[3]);
        const double gv = a == 0 ? rr[0] : (a == 1 ? rr[1] : (a == 2 ? rr[2] : rr[3]));
        if (lo4) gm(Q.g)[i4(Q, tc, k, a)] = gv;
        double pn[13];
        SFOR(j, 0, 3, { pn[j] = p[j]; });
        SFOR(j, 3, 13, { pn[j] = 0.0; });
        dot2bc<6, 0>(pn[3], pn[4], p, ar[0], ar[1]);
        dotbc<6, 0>(pn[5], p, ar[2]);
        dot2bc<10, 0>(pn[6], pn[7], p, ar[3], ar[4]);
        dot2bc<10, 0>(pn[8], pn[9], p, ar[5], ar[6]);
        dot2bc<13, 0>(pn[10], pn[11], p, ar[7], ar[8]);
        dotbc<13, 0>(pn[12], p, ar[9]);
        SFOR(j, 0, 13, { p[j] = pn[j] + P.W[ext_of(j)] * cur.xk[j]; });
    };
    Bk c0, c1;
    loadb(head - 1, c0);
    int k = head - 1;
    while (k >= 0) {
        loadb(imax(k - 1, 0), c1);
        bodyb(c0, k);
        if (--k < 0) break;
        loadb(imax(k - 1, 0), c0);
        bodyb(c1, k);
        --k;
    }
}

// -DCFN_PROF (development builds only, tools/ipm_phase_prof.py): phase timers of the longest wave
#ifdef CFN_PROF
__device__ unsigned long long g_prof[32];   // [0..7] phases of the longest wave, [8] its total, [9] sum of totals, [10] waves, [16..23] phase sums
#define PROF_T(i) { const unsigned long long now_ = wall_clock64(); pacc[i] += now_ - plast; plast = now_; }
#define PROF_SOLVE(h) { psolves++; pstages += (h); }
#ifdef CFN_PROF_MODE
#define CFN_PROF_MODE_OR(m) CFN_PROF_MODE
#else
#define CFN_PROF_MODE_OR(m) (m)
#endif
#else
#define PROF_T(i)
#define PROF_SOLVE(h)
#endif
// One wave = four compacted constrained instances.  MODE 0: everything (active-set solves if
// P.active_set, then the interior point for the rows that did not settle); MODE 1: active-set
// solves only -- rows that settle and pass the tail check are finished and flagged in P.done, the
// others are left untouched for a MODE 2 launch; MODE 2: interior point for the rows without flag;
// MODE 3: MODE 1 for the rows the commit kernel flagged (P.done = 2: settled, but a tail input left the box --
// solve again over the longer head it wrote to P.head); MODE 4: the active-set solves of MODE 1 only -- no
// roll-out: k_ascommit adds the delta to the start solve's candidate in a kernel of its own (deeper prefetch
// than this kernel's registers allow: for small fleets, where the roll-out is a latency chain).
// vb: index of the compact block (group of four list slots) this call works on -- the workgroup index, or the running
// index of a grid-stride loop (k_ipm_rest: a small fixed grid instead of one mostly idle workgroup per four instances)
// CST (fused start solve, Params.fused = 1): the instance's (A, B, b) of ALL stages are already in the wave's compact blocks
// (k_linearise_clist wrote them there; the home blocks hold none) -- nothing to gather but the 4-vectors and the
// checkpoint, and the roll-out reads the compact copy over the whole horizon.
// Fall-back rows start on the full horizon when their classified head is at least this long (1: always; 0: never -- the
// behaviour up to round 5; A/B builds: -DCFN_REST_FULL_HEAD=...)
#ifndef CFN_REDO_WARM
#define CFN_REDO_WARM 1
#endif
#ifndef CFN_REST_FULL_HEAD
#define CFN_REST_FULL_HEAD 1
#endif
template <int MODE, bool SBOX = false, bool CST = false, bool QT = false>
__device__ __forceinline__ void qp_wave(const Params& P, double (*wtile)[WT_TILE], double (*btile)[64], const int vb,
                                        const double* qtab = nullptr) {
#ifdef CFN_PROF
    unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, plast = wall_clock64();
    const unsigned long long pstart = plast;
    unsigned long long psolves = 0, pstages = 0;
#endif
    // MODE 2 works on the list k_ipm_list compacted from the rows the active-set kernels left (P.ilist2, count in
    // P.nipm[NI_LISTED]): four fall-back rows per wave instead of one row in each of the waves they were scattered over
    const bool listed = MODE == 2 && P.ipm_listed;   // (small fleets skip k_ipm_list: the rows stay where k_as had them)
    // (MODE 4 beside the dense kernel: only the rows with heads of more than 16 stages, the first P.nipm[NI_LONG16] of the list)
    // (MODE 2 / 3 behind a split forward sweep: the late rows its second part appended -- P.nipm[NI_LATE] of them -- belong to the list)
    const int nipm = gm(P.nipm)[listed ? NI_LISTED : ((MODE == 4 && P.as_dense) ? NI_LONG16 : 0)] +
                     (((MODE == 2 && !listed) || MODE == 3) && P.fwd_split ? gm(P.nipm)[NI_LATE] : 0);
    const int slot_lo = (MODE == 4 && P.as_dense && P.as_range == 1) ? gm(P.nipm)[NI_LONG24] : 0;   // (first list slot of this launch)
    // SPARSE (active-set kernels, short lists): ONE list slot per wave (row 0; rows 1..3 idle) while the constrained rows
    // number fewer than the SIMDs -- every row then sweeps its own head, restarts at its own stage and stops after its own
    // last solve instead of following the slowest of four wave-mates, and the kernel lasts as long as its hardest ROW.
    // The compact slot of list slot c is c in both modes (row c & 3 of compact block c >> 2), so kernels of either
    // mode read each other's results.
    // (round 6, measured: the interior-point fall-back gains nothing from one row per wave -- kicks x 2: 5.83 -> 5.75 ms for ~230 rows;
    //  the launch lasts as long as its slowest ROW, 26 - 31 iterations of 4 sweeps x 50 stages at ~1 us, whoever shares its wave)
    const bool sparse = (MODE == 1 || MODE == 3 || MODE == 4) && nipm <= P.as_sparse_max && nipm <= (int)gridDim.x;
    const int slot = sparse ? vb : vb * 4 + (threadIdx.x >> 4);
    if ((sparse ? vb : vb * 4) >= nipm || (sparse ? vb : vb * 4 + 3) < slot_lo) return;  // wave-uniform: no work for this wave
    bool has = slot < nipm && slot >= slot_lo && (!sparse || (threadIdx.x >> 4) == 0);
    const int inst0 = has ? gm(listed ? P.ilist2 : P.ilist)[imin(slot, nipm - 1)] : 0;
    constexpr bool AS_ONLY = MODE == 1 || MODE == 3 || MODE == 4;
    constexpr bool NO_ROLL = MODE == 4;   // solves only: roll-out, tail check and publication are left to k_ascommit
    if (MODE == 2 || MODE == 3) {   // MODE 2: rows left for the interior point (done = 0); MODE 3: rows the
        has = has && gm(P.done)[inst0] == (MODE == 2 ? 0 : 2);   // commit kernel sent back for a longer head (done = 2)
        if (!__any(has)) return;
    }
    const Lane t = lane_indirect(P, inst0, has);
    double* wt = wtile[t.row];
    double* sb = btile[t.row];
    const int N = P.N;
    // wave-uniform head = largest head class among the four rows
    int head = t.valid ? gm(P.head)[t.inst] : 0;
    head = max(head, __shfl_xor(head, 16));
    head = max(head, __shfl_xor(head, 32));
    // Fall-back rows (MODE 2: the active set did not settle, or was skipped) run their interior point on the FULL horizon -- the
    // restatement's own algorithm: 99 % of them end there anyway, after a first run over the classified head whose tail
    // check fails (16 of the 36 iterations of the launch's longest wave at kicks x 2, and the launch lasts as long as that
    // wave: 5.7 -> 4.7 ms there, 7.2 -> 6.1 ms at kicks x 3; a small fleet's step with one such row 5.0 -> 3.6 ms;
    // profiles/r06_notes.md section 12).  No tail, hence no tail check and no second attempt.
    if (MODE == 2 && CFN_REST_FULL_HEAD > 0 && head >= CFN_REST_FULL_HEAD) head = N;
    int chk = -1;
    SFOR(c, 0, N_CHK, { if (head == chk_stage(c) && head < N) chk = c; });
    const double viol = t.valid ? gm(P.viol)[t.inst] : 0.0;
    const bool infeasible = t.valid && (viol > 0.0);
    // rows whose unconstrained minimiser lies more than as_skip_viol box widths outside the box do not try the active-set
    // iteration (it settles on 15 % of them beyond 4 widths, 1-2 % beyond 8: measured) -- straight to the interior point
    // (per-stage boxes: never skipped -- a pinned input, lb = ub, is an equality only the active-set solves can hold, and the
    //  scalar box width is no measure of "far outside" when the boxes differ from stage to stage)
    const bool try_as = infeasible && !(!SBOX && P.as_skip_viol > 0.0 && viol > P.as_skip_viol * (P.u_max - P.u_min));
    // All interior-point sweeps run on a COMPACT copy of the head stages (row r of this wave =
    // slot r of compact block blockIdx.x): the instance's own blocks are interleaved with three
    // unrelated instances, which would waste 3/4 of every cache line on every sweep of every
    // iteration.  The start solve's gains / inputs in P stay untouched until the QP is accepted.
    Lane tc = t;
    if (sparse) {   // row 0: compact slot `slot`; idle rows: parked on the spare compact block
        tc.inst = t.row == 0 ? slot : P.NW * 4 + t.row;
        tc.wave = tc.inst >> 2; tc.q = tc.inst & 3;
    } else {
        tc.wave = vb; tc.q = t.row; tc.inst = vb * 4 + t.row;
        // a row without work is parked on the spare compact block as well: beside the dense kernel (MODE 4, P.as_dense) the list
        // slots behind this kernel's share belong to rows k_as_dense is working on AT THE SAME TIME -- an idle row sweeping
        // "its" compact slot would overwrite their du / dx
        if (!has) { tc.wave = P.NW; tc.inst = P.NW * 4 + t.row; }
    }
    Params Q = P;
    Q.v4b = 0;   // (compact 4-vectors: instance-major, a row's head contiguous)
    Q.ab16 = 0;  // (compact A, B: [block][stage])
    Q.AR = P.cAR; Q.BR = P.cBR; Q.KR = P.cKR; Q.Sinv = P.cSinv; Q.d = P.cd; Q.Pchk = P.cPchk; Q.v = P.cv; Q.uit = P.cuit;
    Q.lbs = P.clbs; Q.ubs = P.cubs;
    const size_t cbase = (size_t)tc.inst * N * 4;  // compact 4-vectors of this row
    const Lane& t_ = t; const Lane& tc_ = tc;
    auto gather = [&](int hd, int ck) {
        const Lane t = lane_opaque(t_), tc = lane_opaque(tc_);   // (this phase's addresses stay inside it)
        // four stages per batch, loads first (the source lines are cold: one HBM round trip each)
        for (int k0 = 0; k0 < hd; k0 += 4) {
            double ar[4][10], br[4][4], vv[4], uu[4], blo[4], bhi[4];
            SFOR(j, 0, 4, {
                const int k = imin(k0 + j, hd - 1);
                if (!CST) {
                    ld_ar(blkab(P, P.AR, t, k, SZ_A), t, ar[j]);
                    ld_rows4(blkab(P, P.BR, t, k, SZ_B), t, br[j]);
                }
                vv[j] = gm(P.v)[i4(P, t, k, t.L & 3)];
                uu[j] = gm(P.uit)[i4(P, t, k, t.L & 3)];
                box_at<SBOX>(P, i4(P, t, k, t.L & 3), blo[j], bhi[j]);
            });
            SFOR(j, 0, 4, {
                const int k = k0 + j;
                if (k < hd) {
                    if (!CST) {
                        gdouble* ca = blkab(Q, Q.AR, tc, k, SZ_A);
                        SFOR(sl, 0, 10, { if (t.L < ar_n(sl)) ca[4 * ar_pre(sl) + tc.q * ar_n(sl) + t.L] = ar[j][sl]; });
                        gdouble* cb = blkab(Q, Q.BR, tc, k, SZ_B);
                        SFOR(a, 0, 4, { if (t.L < 13) cb[(a * 4 + tc.q) * 13 + t.L] = br[j][a]; });
                    }
                    if (t.L < 4) {
                        gm(Q.v)[i4(Q, tc, k, t.L)] = vv[j];
                        gm(Q.uit)[i4(Q, tc, k, t.L)] = uu[j];
                        if (SBOX) { gm(Q.lbs)[i4(Q, tc, k, t.L)] = blo[j]; gm(Q.ubs)[i4(Q, tc, k, t.L)] = bhi[j]; }
                    }
                }
            });
        }
        if (ck >= 0) {
            const gdouble* pc = gm(P.Pchk) + ((size_t)t.wave * N_CHK + ck) * SZ_PP;   // home: packed; compact copy: full
            gdouble* qc = gm(Q.Pchk) + ((size_t)tc.wave * N_CHK + ck) * SZ_P;
            double pv[13];
            SFOR(j, 0, 13, { pv[j] = pc[pchk_at(j, t.q, imin(t.L, 12))]; });
            SFOR(j, 0, 13, { if (t.L < 13) qc[(j * 4 + tc.q) * 13 + t.L] = pv[j]; });
        }
    };
    RowIPM R;
    bool accepted = !AS_ONLY;   // MODE 1: only rows finished by the active-set solve are final

    int prev_n = 0;   // inputs of the previous attempt's head (their final classes are still in the compact Q.tu)
    int as_total = 0; // active-set solves of this row over all attempts (what cfnmpc_get_stats reports: the solves it cost)
    for (int attempt = 0; attempt < 3; attempt++) {
        R.iters = 0; R.status = 0; R.res = 0.0; R.mu = 0.0; R.act = false;
        gather(head, chk);
        PROF_T(0)
        // ---- primal-dual active-set solves (exact when the classification becomes stationary)
        bool as_done = false;
        int as_iters = 0;
        if (MODE != 2 && P.active_set) {
            // initial classification from the unconstrained minimiser; WARM (cfnmpc_opts.as_warm): a row whose previous RTI step
            // ended in a settled active-set solve starts from the union of that solve's final set and today's violations (the
            // reference never shifts its iterate, acados_mpc.cpp:581-611, so the classes are taken stage for stage).  Any start
            // ends in the same place: a stationary classification is the exact solution.
            // AGAIN (a further attempt of the active-set kernels: the settled solution's tail left the box, the head got longer): the
            // stages of the previous head start from the classes that attempt ENDED with -- the stationary set of the shorter
            // problem, all but identical to the longer one's -- instead of from the unconstrained minimiser's violations: one or
            // two solves instead of the row's four or five all over again (CFN_REDO_WARM; profiles/r06_notes.md section 14).
            const bool warm = P.as_warm && t.valid && gm(P.wvalid)[t.inst] != 0;
            const gbyte* wc = gm(P.wcls) + (size_t)(warm ? t.inst : 0) * N * 4;
            const int n_again = (CFN_REDO_WARM && AS_ONLY && attempt > 0) ? prev_n : 0;
            for (int e0 = t.L; e0 < head * 4; e0 += 64) {
                double uk[4], vv[4], blo[4], bhi[4], pcl[4];
                int wprev[4];
                SFOR(j, 0, 4, {
                    const size_t idx = cbase + imin(e0 + 16 * j, head * 4 - 1);
                    uk[j] = gm(Q.uit)[idx];
                    vv[j] = gm(Q.v)[idx];
                    box_at<SBOX>(Q, idx, blo[j], bhi[j]);
                    wprev[j] = warm ? (int)wc[imin(e0 + 16 * j, head * 4 - 1)] : 0;
                    pcl[j] = n_again > 0 ? gm(Q.tu)[cbase + imin(e0 + 16 * j, n_again - 1)] : 0.0;
                });
                SFOR(j, 0, 4, {
                    const int e = e0 + 16 * j;
                    if (e < head * 4) {
                        const double lb = blo[j] - uk[j], ub = bhi[j] - uk[j];
                        double cls = vv[j] < lb ? 1.0 : (vv[j] > ub ? 2.0 : 0.0);
                        if (cls == 0.0 && wprev[j] != 0) cls = (double)wprev[j];
                        if (e < n_again) cls = pcl[j];
                        if (SBOX && !(blo[j] < bhi[j])) cls = 1.0;   // lb = ub: fixed from the start
                        gm(Q.tu)[cbase + e] = cls;
                        gm(Q.tl)[cbase + e] = cls == 1.0 ? lb - vv[j] : (cls == 2.0 ? ub - vv[j] : 0.0);
                    }
                });
            }
            bool as_ok = true;
            int kstart = head - 1;
            const bool any_try = __any(try_as);
            for (int it = 1; any_try && it <= AS_MAX_SOLVES; it++) {
                PROF_T(1)
                as_ok = sweep_factor_as<QT>(Q, lane_opaque(tc), head, chk, kstart, wt, sb, qtab) && as_ok;
                PROF_T(2)
                PROF_SOLVE(kstart + 1)
                int jw = sweep_forward_as<SBOX, NO_ROLL>(Q, lane_opaque(tc), head);
                const bool changed = jw >= 0;
                jw = max(jw, __shfl_xor(jw, 16));
                jw = max(jw, __shfl_xor(jw, 32));
                kstart = jw >= 0 ? as_restart_mono(jw, head, as_pg_shift(head)) : head - 1;   // restart point of the next factorisation (wave-uniform)
                PROF_T(3)
                const bool fine = row_min(as_ok ? 1.0 : 0.0) > 0.0;
                if (try_as && !as_done && !changed && fine) { as_done = true; as_iters = it; }
                if (!__any(try_as && !as_done && fine)) break;
            }
            if (P.as_warm && t.valid) {   // what the next RTI step of this row may start from
                if (as_done) {
                    gbyte* wo = gm(P.wcls) + (size_t)t.inst * N * 4;
                    for (int e0 = t.L; e0 < N * 4; e0 += 64) {
                        double cl[4];
                        SFOR(j, 0, 4, { cl[j] = gm(Q.tu)[cbase + imin(e0 + 16 * j, head * 4 - 1)]; });
                        SFOR(j, 0, 4, { const int e = e0 + 16 * j; if (e < N * 4) wo[e] = e < head * 4 ? (unsigned char)(int)cl[j] : (unsigned char)0; });
                    }
                }
                if (t.L == 0) gm(P.wvalid)[t.inst] = as_done ? 1 : 0;
            }
            if (NO_ROLL) {   // hand over to the commit kernel: settled flag, solve count, the head the solves covered
                if (t.L == 0 && t.valid) {
                    gm(P.asst)[tc.inst] = as_done ? 1 : 0;
                    gm(P.head)[t.inst] = head;
                    if (as_done) gm(P.iters)[t.inst] = as_iters;
                }
                return;
            }
            if (as_done) {   // accepted: inputs of the head = v0 + du
                for (int e0 = t.L; e0 < head * 4; e0 += 64) {
                    double vv[4], du[4];
                    SFOR(j, 0, 4, {
                        const size_t idx = cbase + imin(e0 + 16 * j, head * 4 - 1);
                        vv[j] = gm(Q.v)[idx];
                        du[j] = gm(Q.dva)[idx];
                    });
                    SFOR(j, 0, 4, { if (e0 + 16 * j < head * 4) gm(Q.v)[cbase + e0 + 16 * j] = vv[j] + du[j]; });
                }
            }
        }
        if (as_done) { as_total += as_iters; R.status = 0; R.iters = as_total; }
        if constexpr (!AS_ONLY) {
        const bool start_ipm = infeasible && !as_done;
        // Clipped start for rows whose unconstrained minimiser lies more than clip_viol box widths outside the box
        // (vehicles far from their iterate's trajectory: the infeasible start below spends 30 - 60 iterations there)
        // (per-stage boxes: "box widths" = the widest box of the row's head stages)
        double wref = P.u_max - P.u_min;
        if (SBOX && __any(start_ipm)) {
            double wm = 0.0;
            for (int e0 = t.L; e0 < head * 4; e0 += 64) {
                double blo[4], bhi[4];
                SFOR(j, 0, 4, { box_at<SBOX>(Q, cbase + imin(e0 + 16 * j, head * 4 - 1), blo[j], bhi[j]); });
                SFOR(j, 0, 4, { wm = fmax(wm, bhi[j] - blo[j]); });
            }
            wref = row_max(wm);
        }
        const bool clip = start_ipm && P.clip_viol > 0.0 && viol > P.clip_viol * wref;
        double mu0c = P.lam0_min;
        if (__any(clip)) {
            for (int e0 = t.L; e0 < head * 4; e0 += 64) {   // v <- clipped into the box, dv -> Q.dva (0 for the other rows)
                double uk[4], vv[4], blo[4], bhi[4];
                SFOR(j, 0, 4, {
                    const size_t idx = cbase + imin(e0 + 16 * j, head * 4 - 1);
                    uk[j] = gm(Q.uit)[idx];
                    vv[j] = gm(Q.v)[idx];
                    box_at<SBOX>(Q, idx, blo[j], bhi[j]);
                });
                SFOR(j, 0, 4, {
                    const int e = e0 + 16 * j;
                    if (e < head * 4) {
                        const double lb = blo[j] - uk[j], ub = bhi[j] - uk[j], w = ub - lb;
                        const double vc = fmin(fmax(vv[j], lb + P.clip_margin * w), ub - P.clip_margin * w);
                        gm(Q.dva)[cbase + e] = clip ? vc - vv[j] : 0.0;
                        if (clip) gm(Q.v)[cbase + e] = vc;
                    }
                });
            }
            sweep_clip_gradient(P, Q, tc, head, chk);
            double acc = 0.0;
            for (int e0 = t.L; e0 < head * 4; e0 += 64) {
                double uk[4], vv[4], gg[4], blo[4], bhi[4];
                SFOR(j, 0, 4, {
                    const size_t idx = cbase + imin(e0 + 16 * j, head * 4 - 1);
                    uk[j] = gm(Q.uit)[idx];
                    vv[j] = gm(Q.v)[idx];
                    gg[j] = gm(Q.g)[idx];
                    box_at<SBOX>(Q, idx, blo[j], bhi[j]);
                });
                SFOR(j, 0, 4, {
                    if (e0 + 16 * j < head * 4) acc += fabs(gg[j]) * fmin(vv[j] - (blo[j] - uk[j]), (bhi[j] - uk[j]) - vv[j]);
                });
            }
            mu0c = fmax(P.lam0_min, P.mu0_scale * row_sum(acc) / (4.0 * head));
        }
        if (start_ipm) {
            // ---- shift slacks / multipliers positive; residuals of the start; first R^, g
            const double mu0 = clip ? mu0c : fmax(P.mu0_scale * viol, P.lam0_min);
            double mu = 0.0, res = 0.0;
            for (int e0 = t.L; e0 < head * 4; e0 += 64) {
                double uk[4], vv[4], gg[4], blo[4], bhi[4];
                SFOR(j, 0, 4, {
                    const size_t idx = cbase + imin(e0 + 16 * j, head * 4 - 1);
                    uk[j] = gm(Q.uit)[idx];
                    vv[j] = gm(Q.v)[idx];
                    gg[j] = clip ? gm(Q.g)[idx] : 0.0;
                    box_at<SBOX>(Q, idx, blo[j], bhi[j]);
                });
                SFOR(j, 0, 4, {
                    const int e = e0 + 16 * j;
                    if (e < head * 4) {
                        const size_t idx = cbase + e;
                        const double v = vv[j];
                        const double lb = blo[j] - uk[j], ub = bhi[j] - uk[j];
                        // clipped rows: exact slacks, multipliers absorb the gradient; others: slacks floored at thr0
                        // (a pinned input, lb = ub, has no interior: its slacks take the floor in either start, so that the
                        //  iteration stays finite and such a row ends at the iteration cap -- status 2 -- instead of NaN)
                        const bool pinned = SBOX && !(ub > lb);
                        const double tl = (clip && !pinned) ? v - lb : fmax(v - lb, P.thr0), tu = (clip && !pinned) ? ub - v : fmax(ub - v, P.thr0);
                        const double itl = rcp_nr(tl), itu = rcp_nr(tu);
                        const double ll = fmax(gg[j], 0.0) + mu0 * itl, lu = fmax(-gg[j], 0.0) + mu0 * itu;
                        const double rg = gg[j] - ll + lu;
                        gm(Q.tl)[idx] = tl; gm(Q.tu)[idx] = tu; gm(Q.ll)[idx] = ll; gm(Q.lu)[idx] = lu; gm(Q.rg)[idx] = rg;
                        const double rl = v - lb - tl, ru = ub - v - tu;
                        const double Dl = ll * itl, Du = lu * itu;
                        gm(Q.Rh)[idx] = t.wu + Dl + Du;
                        gm(Q.g)[idx] = rg + ll + Dl * rl - lu - Du * ru;
                        mu += ll * tl + lu * tu;
                        res = fmax(res, fmax(fmax(ll * tl, lu * tu), fmax(fabs(rg), fmax(fabs(rl), fabs(ru)))));
                    }
                });
            }
            R.mu = row_sum(mu) / (8.0 * head);
            R.res = row_max(res);
            R.act = true;
            R.status = 2;
        }

        PROF_T(1)
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
            PROF_T(1)
            PROF_SOLVE(head)   // (profiling builds: interior-point iterations count like active-set solves)
            const bool fok = sweep_factor<false>(Q, lane_opaque(tc), head, chk, wt, sb);
            PROF_T(2)
            sweep_forward_delta(Q, lane_opaque(tc), head, gm(Q.dva));
            PROF_T(3)
            // affine step length, mu_aff, centering; corrector right-hand side.  Three dependent
            // passes (row reductions in between); a row with at most 64 inputs (head <= 16) keeps
            // its elements in registers across them.
            double smu;
            {
                const int n = head * 4;
                struct Aff { double dtl, dtu, dll, dlu, itl, itu; };
                auto aff = [&](const Elem& el) {
                    Aff f;
                    const double rl = el.v - el.lb - el.tl, ru = el.ub - el.v - el.tu;
                    f.dtl = el.dva + rl; f.dtu = -el.dva + ru;
                    f.itl = rcp_nr(el.tl); f.itu = rcp_nr(el.tu);
                    f.dll = -el.ll - (el.ll * f.itl) * f.dtl; f.dlu = -el.lu - (el.lu * f.itu) * f.dtu;
                    return f;
                };
                double a = 1.0, mu_aff = 0.0;
                auto passA = [&](const Elem& el, const Aff& f) {
                    a = ratio(el.tl, f.dtl, a); a = ratio(el.tu, f.dtu, a);
                    a = ratio(el.ll, f.dll, a); a = ratio(el.lu, f.dlu, a);
                };
                auto passB = [&](const Elem& el, const Aff& f) {
                    mu_aff += (el.ll + a * f.dll) * (el.tl + a * f.dtl) + (el.lu + a * f.dlu) * (el.tu + a * f.dtu);
                };
                auto passC = [&](int e, const Elem& el, const Aff& f) {
                    const double cl = f.dll * f.dtl, cu = f.dlu * f.dtu;
                    gm(Q.g)[cbase + e] = (cl - smu) * f.itl - (cu - smu) * f.itu;
                };
                auto centre = [&]() {
                    mu_aff = row_sum(mu_aff) / (8.0 * head);
                    const double sr = mu_aff * rcp_nr(R.mu);
                    smu = sr * sr * sr * R.mu;
                };
                if (n <= 64) {
                    Elem d[4];
                    Aff f[4];
                    SFOR(j, 0, 4, { d[j] = ld_elem<1, SBOX>(Q, cbase + imin(t.L + 16 * j, n - 1)); });
                    SFOR(j, 0, 4, { f[j] = aff(d[j]); if (t.L + 16 * j < n) passA(d[j], f[j]); });
                    a = row_min(a);
                    SFOR(j, 0, 4, { if (t.L + 16 * j < n) passB(d[j], f[j]); });
                    centre();
                    SFOR(j, 0, 4, { if (t.L + 16 * j < n) passC(t.L + 16 * j, d[j], f[j]); });
                } else {
                    elem_pass<1, SBOX>(Q, cbase, n, t.L, [&](int, const Elem& el) { passA(el, aff(el)); });
                    a = row_min(a);
                    elem_pass<1, SBOX>(Q, cbase, n, t.L, [&](int, const Elem& el) { passB(el, aff(el)); });
                    centre();
                    elem_pass<1, SBOX>(Q, cbase, n, t.L, [&](int e, const Elem& el) { passC(e, el, aff(el)); });
                }
            }
            // corrector: re-solve, forward
            PROF_T(4)
            sweep_resolve(Q, lane_opaque(tc), head);
            PROF_T(5)
            sweep_forward_delta(Q, lane_opaque(tc), head, gm(Q.dvc));
            PROF_T(3)
            // step, update, residuals of the new point, next R^ and g (two dependent passes)
            {
                const int n = head * 4;
                struct Stp { double dv, dtl, dtu, dll, dlu; };
                auto stp = [&](const Elem& el) {
                    Stp f;
                    f.dv = el.dva + el.dvc;
                    const double rl = el.v - el.lb - el.tl, ru = el.ub - el.v - el.tu;
                    const double dtla = el.dva + rl, dtua = -el.dva + ru;
                    const double itl = rcp_nr(el.tl), itu = rcp_nr(el.tu);
                    const double Dl = el.ll * itl, Du = el.lu * itu;
                    const double cl = (-el.ll - Dl * dtla) * dtla, cu = (-el.lu - Du * dtua) * dtua;
                    f.dtl = f.dv + rl; f.dtu = -f.dv + ru;
                    f.dll = (smu - cl) * itl - el.ll - Dl * f.dtl; f.dlu = (smu - cu) * itu - el.lu - Du * f.dtu;
                    return f;
                };
                double a = 1.0, mu = 0.0, res = 0.0;
                auto passD = [&](const Elem& el, const Stp& f) {
                    a = ratio(el.tl, f.dtl, a); a = ratio(el.tu, f.dtu, a);
                    a = ratio(el.ll, f.dll, a); a = ratio(el.lu, f.dlu, a);
                };
                auto passE = [&](int e, const Elem& el, const Stp& f) {
                    const size_t idx = cbase + e;
                    const double v = el.v + a * f.dv, tl = el.tl + a * f.dtl, tu = el.tu + a * f.dtu;
                    const double ll = el.ll + a * f.dll, lu = el.lu + a * f.dlu, rg = el.rg * (1.0 - a);
                    const double rln = v - el.lb - tl, run = el.ub - v - tu;
                    const double Dl = ll * rcp_nr(tl), Du = lu * rcp_nr(tu);
                    if (R.act) {
                        gm(Q.v)[idx] = v; gm(Q.tl)[idx] = tl; gm(Q.tu)[idx] = tu; gm(Q.ll)[idx] = ll; gm(Q.lu)[idx] = lu; gm(Q.rg)[idx] = rg;
                        gm(Q.Rh)[idx] = t.wu + Dl + Du;
                        gm(Q.g)[idx] = rg + ll + Dl * rln - lu - Du * run;
                    }
                    mu += ll * tl + lu * tu;
                    res = fmax(res, fmax(fmax(ll * tl, lu * tu), fmax(fabs(rg), fmax(fabs(rln), fabs(run)))));
                };
                if (n <= 64) {
                    Elem d[4];
                    Stp f[4];
                    SFOR(j, 0, 4, { d[j] = ld_elem<2, SBOX>(Q, cbase + imin(t.L + 16 * j, n - 1)); });
                    SFOR(j, 0, 4, { f[j] = stp(d[j]); if (t.L + 16 * j < n) passD(d[j], f[j]); });
                    a = fmin(1.0, P.tau * row_min(a));
                    SFOR(j, 0, 4, { if (t.L + 16 * j < n) passE(t.L + 16 * j, d[j], f[j]); });
                } else {
                    elem_pass<2, SBOX>(Q, cbase, n, t.L, [&](int, const Elem& el) { passD(el, stp(el)); });
                    a = fmin(1.0, P.tau * row_min(a));
                    elem_pass<2, SBOX>(Q, cbase, n, t.L, [&](int e, const Elem& el) { passE(e, el, stp(el)); });
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
        }   // !AS_ONLY

        PROF_T(4)
        // ---- expand: dynamics-exact roll-out; head stages use the QP inputs, tail stages the
        //      unconstrained feedback law of the start solve, whose inputs must stay inside the box
        int kviol = -1;  // last tail stage whose feedback input leaves the box
        {
            // the candidate iterate (old + step) goes straight into the new iterate buffers: a row that
            // is not accepted is overwritten again (retry / interior-point launch) or restored (keep_row)
            const Lane t = lane_opaque(t_), tc = lane_opaque(tc_);   // (the roll-out's addresses stay inside it)
            double xbcur = ld13(blk(P.xit, t, N + 1, 0, SZ_V13), t);
            double x = ld13(blk(P.x0, t, 1, 0, SZ_V13), t) - xbcur;
            // head stages take A, B from the wave's compact copy (the home blocks are interleaved with the three wave-mates:
            // three quarters of every cache line foreign) and need no gain; b and the tail come from the home blocks
            auto load_roll = [&](int k, FwdIn<true>& in) {
                if (CST) {   // everything but the start solve's gains from the compact copy
                    ld_ar_raw(blkab(Q, Q.AR, tc, k, SZ_A), tc, in.ar);
                    ld_rows4_raw(blkab(Q, Q.BR, tc, k, SZ_B), tc, in.br);
                    in.bv = ld13_raw(blk(P.cbv, tc, N, k, SZ_V13), tc);
                    if (k >= head) {
                        ld_cols4_raw(blk(P.KR, t, N, k, SZ_K), t, in.kr);
                        in.d = gm(P.d)[i4(P, t, k, t.L & 3)];
                    }
                } else if (k < head) {
                    ld_ar_raw(blkab(Q, Q.AR, tc, k, SZ_A), tc, in.ar);
                    ld_rows4_raw(blkab(Q, Q.BR, tc, k, SZ_B), tc, in.br);
                    in.bv = ld13_raw(blkab(P, P.b, t, k, SZ_V13), t);
                } else {
                    load_fwd<true>(P, t, k, in);
                }
            };
            // Three rotating stage buffers (the stage's 4-vectors and state travel with it): the loads of stage k + 2 are issued
            // before the arithmetic of stage k.  The tail stages come from the HOME blocks (cold lines, a quarter of each the row's
            // own); one stage ahead every tail stage waited a whole HBM round trip -- 3.3 us per stage, 54 % of the mean wave of
            // k_as at 65 536 instances.  The roll-out runs after the solves: their registers are free here.
            struct RollIn { FwdIn<true> f; double v, u, xb, blo, bhi; };
            auto load_stage_roll = [&](int k, RollIn& in) {   // everything stage k reads (k < N)
                load_roll(k, in.f);
                in.v = gm(Q.v)[i4(Q, tc, imin(k, head - 1), t.L & 3)];
                in.u = gm(P.uit)[i4(P, t, k, t.L & 3)];
                box_at<SBOX>(P, i4(P, t, k, t.L & 3), in.blo, in.bhi);
                in.xb = ld13_raw(blk(P.xit, t, N + 1, k + 1, SZ_V13), t);   // x_{k+1} of the old iterate (stored by st13: lanes < 13)
            };
            auto body_roll = [&](const RollIn& cur, int k) {
                st13(blk(P.xitn, t, N + 1, k, SZ_V13), t, xbcur + x);
                double v;
                if (k < head) {
                    v = t.L < 4 ? cur.v : 0.0;
                } else {
                    v = feedback<true>(t, cur.f, x);
                    if (t.L < 4 && !((v >= cur.blo - cur.u) && (v <= cur.bhi - cur.u))) kviol = k;
                }
                // candidate inputs of the whole horizon (P.v keeps the unconstrained minimiser)
                if (t.L < 4) gm(P.uitn)[i4(P, t, k, t.L)] = cur.u + v;
                double vr[4];
                SFOR(a, 0, 4, { vr[a] = bc<a>(v); });
                x = propagate<true>(t, cur.f, x, vr);
                xbcur = cur.xb;
            };
            constexpr int RD = CFN_ROLL_DEPTH;   // rotating buffers: stage k + RD - 1 is requested before stage k is computed
            RollIn rb[RD];
            SFOR(j, 0, RD - 1, { load_stage_roll(imin(j, N - 1), rb[j]); });
            int k = 0;
            while (k < N) {
                SFOR(j, 0, RD, {
                    if (k < N) {
                        load_stage_roll(imin(k + RD - 1, N - 1), rb[(j + RD - 1) % RD]);
                        body_roll(rb[j], k);
                        ++k;
                    }
                });
            }
            st13(blk(P.xitn, t, N + 1, N, SZ_V13), t, xbcur + x);
            kviol = (int)row_max((double)kviol);
        }
        PROF_T(6)
        // MODE 1: a row is finished iff its active set settled and its tail stays in the box; a
        // settled row whose tail leaves the box gets a longer head (below), one that did not
        // settle is left to the interior-point kernel
        if (AS_ONLY) accepted = as_done && kviol < 0;
        const bool redo = AS_ONLY ? (t.valid && as_done && kviol >= 0 && head < N)
                                    : (t.valid && R.status != 4 && kviol >= 0 && head < N);
        if (!__any(redo)) break;
#ifdef CFN_PROF
        if (threadIdx.x == 0 && MODE == CFN_PROF_MODE_OR(MODE)) atomicAdd(&g_prof[15], 1ull);   // (waves that go round again)
#endif
        // rare: a tail input left the box -> solve again (whole wave) over the smallest head class
        // that covers the offending stage (+4), the full horizon as the last resort.  Nothing of
        // the start solve was touched (the sweeps work on the compact copy), so just re-gather.
        int want = redo ? kviol + 5 : 0;
        want = max(want, __shfl_xor(want, 16));
        want = max(want, __shfl_xor(want, 32));
        prev_n = head * 4;
        head = attempt == 0 ? max(head_class(P, want), head) : N;
        chk = -1;
        SFOR(c, 0, N_CHK, { if (head == chk_stage(c) && head < N) chk = c; });
    }
    if (AS_ONLY && t.L == 0 && t.valid) gm(P.done)[t.inst] = accepted ? 1 : 0;
    // the monolithic kernel lists the rows it leaves for the interior point itself (k_scatter zeroed the count): no k_ipm_list
    // launch between the two kernels (12 - 50 us on the step's critical path at 65 536 instances).  The order of the list is
    // that of the waves' completion; a row's result does not depend on its slot or its wave-mates.
    if (MODE == 1 && P.ipm_listed && t.L == 0 && t.valid && !accepted) gm(P.ilist2)[atomicAdd(P.nipm + NI_LISTED, 1)] = t.inst;
    if (t.L == 0 && infeasible && accepted) {
        gm(P.status)[t.inst] = R.status;
        gm(P.iters)[t.inst] = R.iters;
        gm(P.res)[t.inst] = R.res;
        gm(P.head)[t.inst] = head;
    }
    // the roll-out already left the new iterate of an accepted row in place; a row whose QP failed
    // keeps its old iterate (MODE 1 leaves the rows it did not finish to the MODE 2 launch)
    keep_row(P, t, !AS_ONLY && infeasible && !(accepted && R.status != 4));
#ifdef CFN_PROF
    PROF_T(7)
#ifndef CFN_PROF_MODE
#define CFN_PROF_MODE MODE
#endif
    if (threadIdx.x == 0 && MODE == CFN_PROF_MODE) {   // (-DCFN_PROF_MODE=2: only the interior-point kernel reports)
        const unsigned long long tot = plast - pstart;
        const unsigned long long old = atomicMax(&g_prof[8], tot);
        if (tot > old) {   // (racy, development aid) phases of the longest wave
            for (int i = 0; i < 8; i++) g_prof[i] = pacc[i];
            g_prof[11] = psolves; g_prof[12] = pstages;
        }
        atomicAdd(&g_prof[13], psolves); atomicAdd(&g_prof[14], pstages);
        atomicAdd(&g_prof[9], tot);
        atomicAdd(&g_prof[10], 1ull);
        for (int i = 0; i < 8; i++) atomicAdd(&g_prof[16 + i], pacc[i]);
    }
#endif
}
// List of the rows the active-set kernels left for the interior point (P.done = 0), in slot order (deterministic):
// one workgroup, every thread scans a run of consecutive slots.
__global__ __launch_bounds__(1024) void k_ipm_list(Params P) {
    __shared__ int cnt[1024];
    const int tid = threadIdx.x;
    const int n = gm(P.nipm)[0] + (P.fwd_split ? gm(P.nipm)[NI_LATE] : 0);   // (+ the late rows of a split forward sweep)
    const int chunk = (n + 1023) / 1024;
    const int lo = tid * chunk, hi = min(lo + chunk, n);
    int c = 0;
    // batches of eight slots with all loads of a batch in flight together (two memory round trips per batch, not 16);
    // the usual list (a few thousand rows) is one batch, kept in registers for the second pass
    int inst[8];
    bool left[8];
    for (int i0 = lo; i0 < hi; i0 += 8) {
        SFOR(j, 0, 8, { inst[j] = gm(P.ilist)[imin(i0 + j, n - 1)]; });
        SFOR(j, 0, 8, { left[j] = (i0 + j < hi) && gm(P.done)[inst[j]] == 0; });
        SFOR(j, 0, 8, { c += left[j] ? 1 : 0; });
    }
    // inclusive scan over the 1024 threads: shuffles inside the wave, the 16 wave totals through LDS (two barriers)
    int incl = c;
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if ((tid & 63) >= off) incl += v;
    }
    if ((tid & 63) == 63) cnt[tid >> 6] = incl;
    __syncthreads();
    if (tid < 64) {
        int w = tid < 16 ? cnt[tid] : 0;
        for (int off = 1; off < 16; off <<= 1) {
            const int v = __shfl_up(w, off);
            if (tid >= off) w += v;
        }
        if (tid < 16) cnt[16 + tid] = w;     // inclusive totals of the waves
    }
    __syncthreads();
    const int wbase = (tid >> 6) > 0 ? cnt[16 + (tid >> 6) - 1] : 0;
    if (tid == 1023) gm(P.nipm)[NI_LISTED] = wbase + incl;
    int pos = wbase + incl - c;
    if (c == 0) return;
    if (chunk <= 8) {
        SFOR(j, 0, 8, { if (left[j]) gm(P.ilist2)[pos++] = inst[j]; });
    } else {
        for (int i = lo; i < hi; i++) {
            const int in = gm(P.ilist)[i];
            if (gm(P.done)[in] == 0) gm(P.ilist2)[pos++] = in;
        }
    }
}
// rows the interior-point fall-back has to look at: the compacted fall-back list, or the whole list + the late rows of a
// split forward sweep (qp_wave<2> counts the same way)
__device__ __forceinline__ int ipm_rest_rows(const Params& P) {
    return P.ipm_listed ? gm(P.nipm)[NI_LISTED] : gm(P.nipm)[0] + (P.fwd_split ? gm(P.nipm)[NI_LATE] : 0);
}
__global__ __launch_bounds__(64) void k_ipm(Params P) {       // MODE 0: used when active_set = 0
    __shared__ __attribute__((aligned(16))) double wtile[4][WT_TILE];
    __shared__ double btile[4][64];
    qp_wave<0>(P, wtile, btile, blockIdx.x);
}
KALIGN __global__ __launch_bounds__(64) void k_as(Params P) {        // active-set solves
    __shared__ __attribute__((aligned(16))) double wtile[4][WT_TILE];
    __shared__ double btile[4][64];
    __shared__ __attribute__((aligned(16))) double qtab[16 * QT_ROW];
    qtab_fill(P, qtab);
    __syncthreads();
    qp_wave<1, false, false, true>(P, wtile, btile, blockIdx.x, qtab);
}
__global__ __launch_bounds__(64) void k_ipm_rest(Params P) {  // interior point for what k_as left
    __shared__ __attribute__((aligned(16))) double wtile[4][WT_TILE];
    __shared__ double btile[4][64];
    for (int vb = blockIdx.x; vb * 4 < ipm_rest_rows(P); vb += gridDim.x) qp_wave<2>(P, wtile, btile, vb);
}
// the same three for per-stage input boxes (cfnmpc_set_box_stages)
__global__ __launch_bounds__(64) void k_ipm_sbox(Params P) {
    __shared__ __attribute__((aligned(16))) double wtile[4][WT_TILE];
    __shared__ double btile[4][64];
    qp_wave<0, true>(P, wtile, btile, blockIdx.x);
}
__global__ __launch_bounds__(64) void k_as_sbox(Params P) {
    __shared__ __attribute__((aligned(16))) double wtile[4][WT_TILE];
    __shared__ double btile[4][64];
    qp_wave<1, true>(P, wtile, btile, blockIdx.x);
}
__global__ __launch_bounds__(64) void k_ipm_rest_sbox(Params P) {
    __shared__ __attribute__((aligned(16))) double wtile[4][WT_TILE];
    __shared__ double btile[4][64];
    for (int vb = blockIdx.x; vb * 4 < ipm_rest_rows(P); vb += gridDim.x) qp_wave<2, true>(P, wtile, btile, vb);
}
// the three for the fused start solve (Params.fused = 1: stage blocks only in the compact store)
__global__ __launch_bounds__(64) void k_ipm_cst(Params P) {
    __shared__ __attribute__((aligned(16))) double wtile[4][WT_TILE];
    __shared__ double btile[4][64];
    qp_wave<0, false, true>(P, wtile, btile, blockIdx.x);
}
KALIGN __global__ __launch_bounds__(64) void k_as_cst(Params P) {
    __shared__ __attribute__((aligned(16))) double wtile[4][WT_TILE];
    __shared__ double btile[4][64];
    qp_wave<1, false, true>(P, wtile, btile, blockIdx.x);
}
__global__ __launch_bounds__(64) void k_ipm_rest_cst(Params P) {
    __shared__ __attribute__((aligned(16))) double wtile[4][WT_TILE];
    __shared__ double btile[4][64];
    for (int vb = blockIdx.x; vb * 4 < ipm_rest_rows(P); vb += gridDim.x) qp_wave<2, false, true>(P, wtile, btile, vb);
}
__global__ __launch_bounds__(64) void k_as_solves(Params P) {  // MODE 4: active-set solves, no roll-out
    __shared__ __attribute__((aligned(16))) double wtile[4][WT_TILE];
    __shared__ double btile[4][64];
    __shared__ __attribute__((aligned(16))) double qtab[16 * QT_ROW];
    qtab_fill(P, qtab);
    __syncthreads();
    qp_wave<4, false, false, true>(P, wtile, btile, blockIdx.x, qtab);
}
__global__ __launch_bounds__(64) void k_as_retry(Params P) {  // MODE 3: rows the commit kernel sent back
    __shared__ __attribute__((aligned(16))) double wtile[4][WT_TILE];
    __shared__ double btile[4][64];
    qp_wave<3>(P, wtile, btile, blockIdx.x);
}
#ifdef CFN_DEV   // round 3's scheduling experiments (as_passes -2 / 1..12): development builds only since round 6 (measured slower at every fleet size)
// =============================================================================================
// Level-synchronous active-set passes (cfnmpc_opts.as_pipeline)
// =============================================================================================
// The monolithic k_as keeps four constrained instances in one wave until the slowest of them has
// settled (up to 12 solves + roll-out + retries, 432 registers, one wave per SIMD): the kernel lasts as
// long as its hardest wave and its SIMDs idle behind rows that finished long ago.  Here ONE launch =
// ONE active-set solve (backward factorisation with the fixed inputs + forward sweep with
// re-classification) of every instance that has not settled yet:
//   k_asp<true>  : pass 0 over the compacted list (k_scatter order: head class x difficulty); gathers the
//                  head stages from the instance's home blocks on the way (the backward sweep reads A, B
//                  there and leaves a compact copy);
//   k_asp<false> : pass p >= 1 over the work list the previous pass appended to (atomic append into bins
//                  by restart stage, longest first, so that the four rows of a wave sweep similar
//                  lengths); the last launch loops in-wave over the remaining solves (few rows left);
//   k_ascommit   : rows that settled: new iterate = candidate of the start solve + delta (head stages:
//                  element-wise from the pass's own du / dx; tail: delta rolled through the closed loop
//                  A - B K of the unconstrained feedback law, whose inputs are verified against the box);
//                  a row whose tail leaves the box is handed to k_as_retry (longer head), one that did not
//                  settle to k_ipm_rest -- both normally find nothing to do.
// The kernels hold only the two active-set sweeps (<= 256 registers: two waves per SIMD) and run a
// grid-stride loop over groups of four rows.  Compact store, INSTANCE-CONTIGUOUS (a row may sit in
// any wave from pass to pass): slot c owns field + (c N + k) S, S = 97 (A: [slot][lanes < ar_n]),
// 52 (B: [a][13]), 52 (K, G: [l][4]), 16 (rows of S: [c][a]), 4 (element state, instance-major as before),
// 13 per stage 0..N (dx), 14 x 13 per saved cost-to-go.  Results do not depend on which rows share a
// wave: every row restarts its factorisation at its own stage with its own saved cost-to-go.
constexpr int AS_NKB = 7, AS_SETS = 3;
// The passes keep the cost-to-go of every AS_PGRAN-th stage only (it is 182 of the ~470 doubles a factor stage
// moves): a later factorisation restarts at the first such stage behind the last change, i.e. repeats up to
// AS_PGRAN - 1 stages more than necessary (identical arithmetic, identical results).
constexpr int AS_PGRAN = 4;
__device__ __forceinline__ int as_restart(int jm, int head) {   // jm = last stage whose class changed
    const int jr = ((jm + AS_PGRAN) / AS_PGRAN) * AS_PGRAN - 1;   // smallest stage >= jm with a saved successor
    return (jr + 1 < head && (jr + 1) / AS_PGRAN < AS_PSAVE) ? jr : head - 1;
}
__host__ __device__ constexpr int as_kbin(int len) {   // len = stages the next factorisation covers
    return len > 32 ? 0 : (len > 24 ? 1 : (len > 16 ? 2 : (len > 12 ? 3 : (len > 8 ? 4 : (len > 4 ? 5 : 6)))));
}
__device__ __forceinline__ void zld_ar_raw(const gdouble* b, const Lane& t, double (&ar)[10]) {
    SFOR(s, 0, 10, { ar[s] = b[ar_pre(s) + imin(t.L, ar_n(s) - 1)]; });
}
__device__ __forceinline__ void zld_ar(const gdouble* b, const Lane& t, double (&ar)[10]) {
    SFOR(s, 0, 10, {
        const double v = b[ar_pre(s) + imin(t.L, ar_n(s) - 1)];
        ar[s] = t.L < ar_n(s) ? v : 0.0;
    });
}
__device__ __forceinline__ void zld_rows4_raw(const gdouble* b, const Lane& t, double (&r)[4]) {
    SFOR(a, 0, 4, { r[a] = b[a * 13 + imin(t.L, 12)]; });
}
__device__ __forceinline__ void zld_rows4(const gdouble* b, const Lane& t, double (&r)[4]) {
    SFOR(a, 0, 4, {
        const double v = b[a * 13 + imin(t.L, 12)];
        r[a] = t.L < 13 ? v : 0.0;
    });
}
struct ZStage {
    StageIn<true> in;
    double v0, uk, c, cls;   // lanes a < 4 (replicated over a = L & 3)
};
// issue the loads of stage k (no use of the values here: the caller consumes them one stage later)
template <bool FIRST>
__device__ __forceinline__ void zload_stage(const Params& P, const Params& Q, const Lane& th, const Lane& tc, const int k,
                                            ZStage& z) {
    const int a = tc.L & 3;
    if (FIRST) {   // from the instance's home blocks (interleaved with its three wave-mates)
        ld_ar_raw(blkab(P, P.AR, th, k, SZ_A), th, z.in.ar);
        ld_rows4_raw(blkab(P, P.BR, th, k, SZ_B), th, z.in.br);
        z.v0 = gm(P.v)[i4(P, th, k, a)];
        z.uk = gm(P.uit)[i4(P, th, k, a)];
        z.c = 0.0; z.cls = 0.0;
    } else {
        const size_t s = (size_t)tc.inst * P.N + k;
        zld_ar_raw(gm(Q.AR) + s * 97, tc, z.in.ar);
        zld_rows4_raw(gm(Q.BR) + s * 52, tc, z.in.br);
        z.c = gm(Q.tl)[i4(Q, tc, k, a)];
        z.cls = gm(Q.tu)[i4(Q, tc, k, a)];
        z.v0 = 0.0; z.uk = 0.0;
    }
}
// Backward sweep of one active-set solve over stages kw .. 0 (kw = the wave's largest restart stage).
// Row state: head (its own head class), chk (checkpoint index of that class, -1: terminal cost),
// join (the stage ITS sweep starts at: head - 1, or the last stage whose class changed in the previous
// solve; -1: the row takes no part).  A row computes from stage kw on (wave-uniform code) but loads
// its cost-to-go when the sweep reaches `join` and stores nothing before that.
template <bool FIRST>
__device__ __forceinline__ bool zsweep_factor(const Params& P, const Params& Q, const Lane& th, const Lane& tc,
                                              const int head, const int chk, const int join, const int kw,
                                              double* wt, double* sb) {
    double Pa[13];
    SFOR(j, 0, 13, { Pa[j] = 0.0; });
    double wq = 0.0;
    SFOR(j, 0, 13, { if (tc.L == j) wq = P.W[ext_of(j)]; });
    const double is13 = tc.L == 13 ? 1.0 : 0.0;
    auto start_row = [&](int k) {   // rows with join == k: cost-to-go of stage k + 1
        if (k != join) return;
        if (k + 1 < head) {         // saved by an earlier solve of this instance
            const gdouble* ps = gm(Q.cPs) + ((size_t)tc.inst * AS_PSAVE + (k + 1) / AS_PGRAN) * 182 + imin(tc.L, 13);
            SFOR(j, 0, 13, { Pa[j] = ps[j * 14]; });
        } else if (chk < 0) {
            SFOR(j, 0, 13, { Pa[j] = (tc.L == j) ? P.WN[ext_of(j)] : 0.0; });
        } else {                    // checkpoint of the unconstrained tail (home block)
            const gdouble* pc = gm(P.Pchk) + ((size_t)th.wave * N_CHK + chk) * SZ_PP;
            SFOR(j, 0, 13, {
                const double v = pc[pchk_at(j, th.q, imin(tc.L, 12))];
                Pa[j] = tc.L < 13 ? v : 0.0;
            });
        }
    };
    bool ok = true;
    auto stage = [&](ZStage& z, int k) {
        if (__any(k == join)) start_row(k);
        const bool act = k <= join;
        const int a = tc.L & 3;
        if (FIRST) {   // initial classification from the unconstrained minimiser
            const double lb = P.u_min - z.uk, ub = P.u_max - z.uk;
            z.cls = z.v0 < lb ? 1.0 : (z.v0 > ub ? 2.0 : 0.0);
            z.c = z.cls == 1.0 ? lb - z.v0 : (z.cls == 2.0 ? ub - z.v0 : 0.0);
            if (act) {     // compact copy of the stage
                const size_t s = (size_t)tc.inst * P.N + k;
                gdouble* ca = gm(Q.AR) + s * 97;
                SFOR(sl, 0, 10, { if (tc.L < ar_n(sl)) ca[ar_pre(sl) + tc.L] = z.in.ar[sl]; });
                gdouble* cb = gm(Q.BR) + s * 52;
                SFOR(aa, 0, 4, { if (tc.L < 13) cb[aa * 13 + tc.L] = z.in.br[aa]; });
                if (tc.L < 4) {
                    const size_t idx = i4(Q, tc, k, tc.L);
                    gm(Q.v)[idx] = z.v0; gm(Q.uit)[idx] = z.uk; gm(Q.tl)[idx] = z.c; gm(Q.tu)[idx] = z.cls;
                }
            }
        }
        const double ra = tc.wu;
        z.in.Rh = z.cls != 0.0 ? AS_BIG * fmax(1.0, ra) : ra;
        z.in.g = 0.0;
        z.in.qv = 0.0;
        const double cm = tc.L < 4 ? z.c : 0.0;
        double bv = 0.0;
        SFOR(aa, 0, 4, { bv += z.in.br[aa] * bc<aa>(cm); });
        z.in.bv = bv;
        (void)a;
        const bool fo = factor_stage<true, true, true>(Q, tc, k, Pa, z.in, wq, is13, wt, sb, act);
        ok = ok && (fo || !act);
        if (act && k > 0 && k % AS_PGRAN == 0 && k / AS_PGRAN < AS_PSAVE && tc.L < 14) {
            gdouble* ps = gm(Q.cPs) + ((size_t)tc.inst * AS_PSAVE + k / AS_PGRAN) * 182 + tc.L;
            SFOR(j, 0, 13, { ps[j * 14] = Pa[j]; });
        }
    };
    ZStage bufA, bufB;
    zload_stage<FIRST>(P, Q, th, tc, kw, bufA);
    int k = kw;
    while (k >= 0) {
        zload_stage<FIRST>(P, Q, th, tc, imax(k - 1, 0), bufB);
        stage(bufA, k);
        if (--k < 0) break;
        zload_stage<FIRST>(P, Q, th, tc, imax(k - 1, 0), bufA);
        stage(bufB, k);
        --k;
    }
    return ok;
}
// Forward sweep of the solve over stages [0, hw) (hw = the wave's largest head): du -> Q.dva, dx -> P.czdx,
// multipliers and re-classification on the way (sweep_forward_as on the compact z layout).  Returns the
// last stage < head of the row in which an input changed its class (-1: none).
__device__ __forceinline__ int zsweep_forward(const Params& P, const Params& Q, const Lane& tc, const int head, const int hw) {
    struct In { double kg[13], ar[10], br[4], d, sr[4], rho, c, cls, v0, uk; };
    const int a = tc.L & 3;
    const bool lo4 = tc.L < 4;
    auto load = [&](int k, In& in) {
        const size_t s = (size_t)tc.inst * P.N + k;
        const gdouble* src = (lo4 ? gm(Q.KR) : gm(Q.cGR)) + s * 52 + a;
        SFOR(l, 0, 13, { in.kg[l] = src[l * 4]; });
        zld_ar(gm(Q.AR) + s * 97, tc, in.ar);
        zld_rows4(gm(Q.BR) + s * 52, tc, in.br);
        const gdouble* sr = gm(Q.cS) + s * 16 + a;
        SFOR(c, 0, 4, { in.sr[c] = sr[c * 4]; });
        const size_t idx = i4(Q, tc, k, a);
        in.d = gm(Q.d)[idx];
        in.rho = gm(Q.crho)[idx];
        in.c = gm(Q.tl)[idx]; in.cls = gm(Q.tu)[idx]; in.v0 = gm(Q.v)[idx]; in.uk = gm(Q.uit)[idx];
    };
    double x = 0.0;
    int jm = -1;
    gdouble* zx = gm(P.czdx) + (size_t)tc.inst * (P.N + 1) * 13 + imin(tc.L, 12);
    auto body = [&](const In& cur, int k) {
        double acc = 0.0;
        dotbc<13, 0>(acc, cur.kg, x);        // lanes 0..3: K[a] dx, lanes 4..7: G[a] dx
        settle(acc);
        double dv = lo4 ? -cur.d - acc : 0.0;
        dv = (lo4 && cur.cls != 0.0) ? cur.c : dv;
        if (lo4) gm(Q.dva)[i4(Q, tc, k, tc.L)] = dv;
        double gd = shift4(acc);             // lane a <- lane a + 4
        double vr[4], fr[4];
        SFOR(c, 0, 4, { vr[c] = bc<c>(dv); });
        const double dfree = cur.cls == 0.0 ? dv : 0.0;
        SFOR(c, 0, 4, { fr[c] = bc<c>(dfree); });
        SFOR(c, 0, 4, { gd += cur.sr[c] * fr[c]; });     // + (B'PB)[a][free] du_free
        if (lo4) {
            const double grad = tc.wu * cur.c + gd + cur.rho;   // multiplier of a fixed input
            const double lb = P.u_min - cur.uk, ub = P.u_max - cur.uk;
            const double vn = cur.v0 + dv;
            double nc;
            if (cur.cls == 0.0) nc = vn < lb ? 1.0 : (vn > ub ? 2.0 : 0.0);
            else if (cur.cls == 1.0) nc = grad > 0.0 ? 1.0 : 0.0;
            else nc = grad < 0.0 ? 2.0 : 0.0;
            jm = (nc != cur.cls && k < head) ? k : jm;
            const size_t idx = i4(Q, tc, k, a);
            gm(Q.tu)[idx] = nc;
            gm(Q.tl)[idx] = nc == 1.0 ? lb - cur.v0 : (nc == 2.0 ? ub - cur.v0 : 0.0);
        }
        double xn = tc.L < 3 ? x : 0.0;
        dotbc<10, 3>(xn, cur.ar, x);
        SFOR(c, 0, 4, { xn += cur.br[c] * vr[c]; });
        x = xn;
        if (tc.L < 13) zx[(size_t)(k + 1) * 13] = x;   // dx_{k+1} of this solve
    };
    In b0, b1;
    load(0, b0);
    int k = 0;
    while (k < hw) {
        load(imin(k + 1, hw - 1), b1);
        body(b0, k);
        if (++k >= hw) break;
        load(imin(k + 1, hw - 1), b0);
        body(b1, k);
        ++k;
    }
    return (int)row_max((double)jm);
}
#endif   // CFN_DEV
__device__ __forceinline__ Params compact_params(const Params& P) {
    Params Q = P;
    Q.v4b = 0;   // (compact 4-vectors: instance-major, a row's head contiguous)
    Q.ab16 = 0;  // (compact A, B: [block][stage])
    Q.AR = P.cAR; Q.BR = P.cBR; Q.KR = P.cKR; Q.Sinv = P.cSinv; Q.d = P.cd; Q.Pchk = P.cPchk; Q.v = P.cv; Q.uit = P.cuit;
    return Q;
}
// One group of four rows of a pass: compact slots, home lanes, per-row sweep state.
#ifdef CFN_DEV
struct AspGroup {
    Lane th, tc;
    int c, inst, head, chk, join;
    bool has, skip;   // skip: the row does not try the active-set iteration (cfnmpc_opts.as_skip_viol)
};
struct AspLists {   // work lists of this pass
    int nwork, pre[AS_NKB], set_in, set_out, cap;
};
template <bool FIRST>
__device__ __forceinline__ AspLists asp_lists(const Params& P, const int pass) {
    AspLists w;
    w.cap = (P.NW + 1) * 4;   // compact slots incl. the four spare ones
    w.set_in = pass % AS_SETS; w.set_out = (pass + 1) % AS_SETS;
    w.nwork = 0;
    if (FIRST) {
        w.nwork = gm(P.nipm)[0];
    } else {
        SFOR(b, 0, AS_NKB, { w.pre[b] = w.nwork; w.nwork += gm(P.ascnt)[w.set_in * AS_NKB + b]; });
    }
    return w;
}
template <bool FIRST>
__device__ __forceinline__ AspGroup asp_group(const Params& P, const AspLists& w, const int g) {
    AspGroup r;
    const int row = threadIdx.x >> 4;
    const int wi = g * 4 + row;
    r.has = wi < w.nwork;
    r.c = w.cap - 4 + row;         // rows without work scribble on a spare slot
    if (r.has) {
        if (FIRST) {
            r.c = wi;
        } else {
            int b = 0, base = 0;
            SFOR(bb, 1, AS_NKB, { if (wi >= w.pre[bb]) { b = bb; base = w.pre[bb]; } });
            r.c = gm(P.aslist)[(size_t)(w.set_in * AS_NKB + b) * w.cap + (wi - base)];
        }
    }
    r.inst = r.has ? gm(P.ilist)[r.c] : 0;
    r.th = lane_indirect(P, r.inst, r.has);
    r.tc = r.th;
    r.tc.inst = r.c; r.tc.wave = r.c >> 2; r.tc.q = r.c & 3;
    r.head = r.has ? gm(P.head)[r.inst] : 0;
    r.chk = -1;
    SFOR(cc, 0, N_CHK, { if (r.head == chk_stage(cc) && r.head < P.N) r.chk = cc; });
    r.join = r.has ? (FIRST ? r.head - 1 : gm(P.askst)[r.c]) : -1;
    r.skip = false;
    if (FIRST && r.has && P.as_skip_viol > 0.0) {
        r.skip = gm(P.viol)[r.inst] > P.as_skip_viol * (P.u_max - P.u_min);
        if (r.skip) r.join = -1;
    }
    return r;
}
__device__ __forceinline__ int wave_max(int v) {   // over the four rows (v is row-uniform)
    v = max(v, __shfl_xor(v, 16));
    return max(v, __shfl_xor(v, 32));
}
// what becomes of a row after a solve: settled -> commit; not positive definite or out of solves -> interior
// point; otherwise it joins the next pass's list, binned by the length of its next factorisation
__device__ __forceinline__ void asp_finish(const Params& P, const AspLists& w, const AspGroup& r, const int solves,
                                           const bool ok, const int jm) {
    if (!r.has || r.tc.L != 0) return;
    if (r.skip) {
        gm(P.asst)[r.c] = 0;
    } else if (ok && jm < 0) {
        gm(P.asst)[r.c] = 1;
        gm(P.iters)[r.inst] = solves;
    } else if (!ok || solves >= AS_MAX_SOLVES) {
        gm(P.asst)[r.c] = 0;
    } else {
        const int join = as_restart(jm, r.head);   // restart stage of the next factorisation
        gm(P.askst)[r.c] = join;
        const int b = as_kbin(join + 1);
        const int pos = atomicAdd(P.ascnt + w.set_out * AS_NKB + b, 1);
        gm(P.aslist)[(size_t)(w.set_out * AS_NKB + b) * w.cap + pos] = r.c;
    }
}
// pass p, backward half: one factorisation per listed row; leaves `ok` per compact slot
template <bool FIRST>
__device__ __forceinline__ void asf_body(const Params& P, const int pass, double (*wtile)[WT_TILE], double (*btile)[64]) {
    const AspLists w = asp_lists<FIRST>(P, pass);
    const Params Q = compact_params(P);
    const int row = threadIdx.x >> 4;
    for (int g = blockIdx.x; g * 4 < w.nwork; g += gridDim.x) {
        const AspGroup r = asp_group<FIRST>(P, w, g);
        const int kw = wave_max(r.join);
        if (kw < 0) continue;
        bool ok = zsweep_factor<FIRST>(P, Q, r.th, r.tc, r.head, r.chk, r.join, kw, wtile[row], btile[row]);
        ok = row_min(ok ? 1.0 : 0.0) > 0.0;
        if (r.has && r.tc.L == 0) gm(P.asok)[r.c] = ok ? 1 : 0;
    }
}
// pass p, forward half: inputs, multipliers, re-classification; builds the next pass's lists
template <bool FIRST>
__device__ __forceinline__ void asw_body(const Params& P, const int pass) {
    const AspLists w = asp_lists<FIRST>(P, pass);
    // (the set the NEXT pass appends to was this pass's predecessor's input: nobody reads it any more)
    if (blockIdx.x == 0 && threadIdx.x < AS_NKB) gm(P.ascnt)[((pass + 2) % AS_SETS) * AS_NKB + threadIdx.x] = 0;
    const Params Q = compact_params(P);
    for (int g = blockIdx.x; g * 4 < w.nwork; g += gridDim.x) {
        const AspGroup r = asp_group<FIRST>(P, w, g);
        const int hw = wave_max(r.skip ? 0 : r.head);
        if (hw <= 0) { asp_finish(P, w, r, pass + 1, false, 0); continue; }
        const int jm = zsweep_forward(P, Q, r.tc, r.skip ? 0 : r.head, hw);
        const bool ok = r.has && gm(P.asok)[r.c] != 0;
        asp_finish(P, w, r, pass + 1, ok, jm);
    }
}
// Several solves in-wave (one wave per SIMD, no spills): FIRST = false -- the remaining solves of the rows still
// unsettled after the single-solve passes (few rows); FIRST = true -- every solve of every constrained instance
// in one launch (as_passes = -2: no level synchronisation at all, the shortest dependent chain -- for small
// fleets, where the SIMDs idle anyway), the first one gathering from the home blocks.
template <bool FIRST>
__device__ __forceinline__ void asp_body(const Params& P, const int pass, const int nsolve, double (*wtile)[WT_TILE],
                                         double (*btile)[64]) {
    const AspLists w = asp_lists<FIRST>(P, pass);
    if (blockIdx.x == 0 && threadIdx.x < AS_NKB) gm(P.ascnt)[((pass + 2) % AS_SETS) * AS_NKB + threadIdx.x] = 0;
    const Params Q = compact_params(P);
    const int row = threadIdx.x >> 4;
    for (int g = blockIdx.x; g * 4 < w.nwork; g += gridDim.x) {
        AspGroup r = asp_group<FIRST>(P, w, g);
        bool active = r.has && !r.skip, ok = true;
        int jm = 0, done_here = 0;
        for (int it = 0; it < nsolve; it++) {
            const int kw = wave_max(active ? r.join : -1), hw = wave_max(active ? r.head : 0);
            if (kw < 0) break;
            bool fo;
            if (FIRST && it == 0) fo = zsweep_factor<true>(P, Q, r.th, r.tc, r.head, r.chk, r.join, kw, wtile[row], btile[row]);
            else fo = zsweep_factor<false>(P, Q, r.th, r.tc, r.head, r.chk, active ? r.join : -1, kw, wtile[row], btile[row]);
            fo = row_min(fo ? 1.0 : 0.0) > 0.0;
            const int j = zsweep_forward(P, Q, r.tc, active ? r.head : 0, hw);
            if (active) {
                done_here++;
                ok = fo; jm = j;
                if (!ok || jm < 0) active = false;
                else r.join = as_restart(jm, r.head);
            }
        }
        asp_finish(P, w, r, pass + done_here, ok, jm);
    }
}
__global__ __launch_bounds__(64, 2) void k_asf_first(Params P) {
    __shared__ __attribute__((aligned(16))) double wtile[4][WT_TILE];
    __shared__ double btile[4][64];
    asf_body<true>(P, 0, wtile, btile);
}
__global__ __launch_bounds__(64, 2) void k_asf(Params P, int pass) {
    __shared__ __attribute__((aligned(16))) double wtile[4][WT_TILE];
    __shared__ double btile[4][64];
    asf_body<false>(P, pass, wtile, btile);
}
__global__ __launch_bounds__(64, 2) void k_asw_first(Params P) { asw_body<true>(P, 0); }
__global__ __launch_bounds__(64, 2) void k_asw(Params P, int pass) { asw_body<false>(P, pass); }
__global__ __launch_bounds__(64) void k_asp(Params P, int pass, int nsolve) {
    __shared__ __attribute__((aligned(16))) double wtile[4][WT_TILE];
    __shared__ double btile[4][64];
    asp_body<false>(P, pass, nsolve, wtile, btile);
}
__global__ __launch_bounds__(64) void k_asp_all(Params P) {
    __shared__ __attribute__((aligned(16))) double wtile[4][WT_TILE];
    __shared__ double btile[4][64];
    asp_body<true>(P, 0, AS_MAX_SOLVES, wtile, btile);
}
// Settled rows: new iterate = candidate (start solve) + delta.  One wave = four consecutive compact slots.
#endif   // CFN_DEV
// DEEP: three rotating stage buffers in the tail loop (one wave per SIMD: small fleets, where the tail is a
// latency chain); otherwise two (two waves per SIMD: large fleets, where it is bound by the home blocks' bytes).
#ifndef CFN_COMMIT_DEPTH
#define CFN_COMMIT_DEPTH 3
#endif
template <bool DEEP>
__device__ __forceinline__ void ascommit_body(const Params& P) {
    const int nipm = gm(P.nipm)[0];
    const int N = P.N;
    const int row = threadIdx.x >> 4;
    const Params Q = compact_params(P);
    for (int g = blockIdx.x; g * 4 < nipm; g += gridDim.x) {
        const int c = g * 4 + row;
        const bool has = c < nipm;
        const int inst = has ? gm(P.ilist)[c] : 0;
        const bool go = has && gm(P.asst)[imin(c, nipm - 1)] == 1;
        const int untried = has && gm(P.asst)[imin(c, nipm - 1)] == 2;   // (k_scatter: head behind the split point -> retry kernel)
        if (!__any(go)) {
            if (has && (threadIdx.x & 15) == 0) gm(P.done)[inst] = untried ? 2 : 0;
            continue;
        }
        const Lane t = lane_indirect(P, inst, go);
        const int head = go ? gm(P.head)[inst] : N;
        const int a = t.L & 3;
        const bool lo4 = t.L < 4;
        int hmin = head;
        hmin = min(hmin, __shfl_xor(hmin, 16)); hmin = min(hmin, __shfl_xor(hmin, 32));
        const int lx = t.q * 13 + imin(t.L, 12);
        const gdouble* zx = gm(P.czdx) + (size_t)c * (N + 1) * 13 + imin(t.L, 12);
        const size_t cb4 = (size_t)c * N * 4 + a;
        double x = 0.0;   // dx_k (lanes 0..12)
        int kviol = -1;
        // (1) stages before the shortest head of the wave: element-wise, nothing read from the home blocks
        //     but the candidate itself (batches of four stages, loads first)
        for (int k0 = 0; k0 < hmin; k0 += 4) {
            double un[4], xc[4], dv[4], zv[4];
            SFOR(j, 0, 4, {
                const int k = imin(k0 + j, hmin - 1);
                un[j] = gm(P.uitn)[i4(P, t, k, a)];
                xc[j] = blk(P.xitn, t, N + 1, k + 1, SZ_V13)[lx];
                dv[j] = gm(Q.dva)[cb4 + (size_t)k * 4];
                zv[j] = zx[(size_t)(k + 1) * 13];
            });
            SFOR(j, 0, 4, {
                const int k = k0 + j;
                if (k < hmin && go) {
                    if (lo4) gm(P.uitn)[i4(P, t, k, t.L)] = un[j] + dv[j];
                    if (t.L < 13) blk(P.xitn, t, N + 1, k + 1, SZ_V13)[lx] = xc[j] + zv[j];
                }
            });
        }
        if (hmin > 0) x = t.L < 13 ? zx[(size_t)hmin * 13] : 0.0;
        // (2) from there on: rows still inside their head keep adding the pass's own du / dx, the others roll
        //     delta through the closed loop of the unconstrained feedback law (K, A, B of the home blocks)
        struct In { double kr[13], ar[10], br[4], un, xc, dv, zx; };
        auto load = [&](int k, In& in) {
            ld_cols4_raw(blk(P.KR, t, N, k, SZ_K), t, in.kr);   // (masked in body())
            ld_ar_raw(blkab(P, P.AR, t, k, SZ_A), t, in.ar);
            ld_rows4_raw(blkab(P, P.BR, t, k, SZ_B), t, in.br);
            in.un = gm(P.uitn)[i4(P, t, k, a)];
            in.xc = blk(P.xitn, t, N + 1, k + 1, SZ_V13)[lx];
            in.dv = gm(Q.dva)[cb4 + (size_t)imin(k, imax(head - 1, 0)) * 4];
            in.zx = zx[(size_t)imin(k + 1, head) * 13];
        };
        auto body = [&](const In& cur, int k) {
            const bool tail = k >= head;
            double acc = 0.0;
            double kr[13], ar[10], br[4];
            mask_cols4(t, cur.kr, kr);
            mask_ar(t, cur.ar, ar);
            mask_rows4(t, cur.br, br);
            dotbc<13, 0>(acc, kr, x);
            double v = lo4 ? -acc : 0.0;
            settle(v);
            v = tail ? v : (lo4 ? cur.dv : 0.0);
            const double un = cur.un + v;
            if (lo4 && go) {
                if (tail && !((un >= P.u_min) && (un <= P.u_max))) kviol = k;
                gm(P.uitn)[i4(P, t, k, t.L)] = un;
            }
            double vr[4];
            SFOR(cc, 0, 4, { vr[cc] = bc<cc>(v); });
            double xn = t.L < 3 ? x : 0.0;
            dotbc<10, 3>(xn, ar, x);
            SFOR(cc, 0, 4, { xn += br[cc] * vr[cc]; });
            x = (k + 1 <= head) ? (t.L < 13 ? cur.zx : 0.0) : xn;
            if (t.L < 13 && go) blk(P.xitn, t, N + 1, k + 1, SZ_V13)[lx] = cur.xc + x;
        };
        if (hmin < N && !DEEP) {
            In b0, b1;
            load(hmin, b0);
            int k = hmin;
            while (k < N) {
                load(imin(k + 1, N - 1), b1);
                body(b0, k);
                if (++k >= N) break;
                load(imin(k + 1, N - 1), b0);
                body(b1, k);
                ++k;
            }
        }
        if (hmin < N && DEEP) {
            // CFN_COMMIT_DEPTH rotating stage buffers: the loads of stage k + DEPTH - 1 are issued before the arithmetic of stage k
            // (one wave per SIMD, the home blocks of a constrained row come from HBM / MALL: ~1 - 2 us per round trip against
            // ~0.15 us of arithmetic per stage)
            constexpr int D = CFN_COMMIT_DEPTH;
            In b[D];
            SFOR(j, 0, D - 1, { load(imin(hmin + j, N - 1), b[j]); });
            int k = hmin;
            while (k < N) {
                SFOR(j, 0, D, {
                    if (k < N) {
                        load(imin(k + D - 1, N - 1), b[(j + D - 1) % D]);
                        body(b[j], k);
                        ++k;
                    }
                });
            }
        }
        kviol = (int)row_max((double)kviol);
        if (has && t.L == 0) {
            if (!go) {
                gm(P.done)[inst] = untried ? 2 : 0;   // did not settle: interior point (not tried yet: retry kernel)
            } else if (kviol < 0) {
                gm(P.done)[inst] = 1;
                gm(P.status)[inst] = 0;
                gm(P.res)[inst] = 0.0;
            } else {                           // tail left the box: again over a longer head (k_as_retry)
                gm(P.done)[inst] = 2;
                gm(P.head)[inst] = max(head_class(P, kviol + 5), head);
            }
        }
    }
}
__global__ __launch_bounds__(64, 2) void k_ascommit(Params P) { ascommit_body<false>(P); }
__global__ __launch_bounds__(64) void k_ascommit1(Params P) { ascommit_body<true>(P); }
#ifdef CFN_PROF
// isolated sweeps on one wave per SIMD (development aid): every wave repeats the sweep `reps` times
__global__ __launch_bounds__(64) void k_bench_sweep(Params P, int head, int reps, int which) {
    __shared__ __attribute__((aligned(16))) double wtile[4][WT_TILE];
    __shared__ double btile[4][64];
    const Lane t = lane_id(P);
    bool ok = true;
    for (int r = 0; r < reps; r++) {
        if (which == 0) ok = sweep_factor<false>(P, t, head, -1, wtile[t.row], btile[t.row]) && ok;
        if (which == 1) sweep_forward_delta(P, t, head, gm(P.dva));
        if (which == 2) sweep_resolve(P, t, head);
        if (which == 3) ok = sweep_factor_as(P, t, head, -1, head - 1, wtile[t.row], btile[t.row]) && ok;
        if (which == 4) ok = sweep_forward_as(P, t, head) < 0 && ok;
    }
    if (!ok && t.L == 77) gm(P.res)[0] = 1.0;
}
float debug_bench_sweep(const Params& P, int waves, int head, int reps, int which) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_bench_sweep, dim3(waves), dim3(64), 0, 0, P, head, 1, which);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_bench_sweep, dim3(waves), dim3(64), 0, 0, P, head, reps, which);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
void debug_prof_read(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 32);
    if (reset) { unsigned long long z[32] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof z); }
}
#endif


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

// State assembly + delay compensation of the reference estimator, batched, one vehicle per lane
// (ESTIMATOR::predictor, acados_estimator.cpp:521-634):
//   meas [B][9] = mocap x y z [m] | onboard roll pitch yaw [deg, as published by the driver]
//                 | gyro rates wx wy wz [rad/s]
//   filt [B][9] = previous position (3), previous two velocity outputs v[k-1] (3), v[k-2] (3);
//                 updated in place (the x/y/z_samples and v*_filter_samples of :370-412)
//   u    [B][4] = latest motor speeds [kRPM] used for the prediction
// Steps: pitch sign flip (:495), deg -> rad, Euler -> quaternion with the reference's sign
// convention and w >= 0 (:327-354) + normalisation (:546), world-velocity low-pass filter
// (:356-368; use_lpf = 0 selects its finite-difference branch), rotation to the body frame
// (:414-440), then one RK4 integration over `delay` in `steps` sub-steps (:573-593).
// Writes x_est (assembled state) and x_pred (delay-compensated state).
__global__ void k_estimate(int B, const double* __restrict__ meas, double* __restrict__ filt,
                           const double* __restrict__ u, double dt, int use_lpf, double delay, int steps,
                           double* __restrict__ x_est, double* __restrict__ x_pred) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const double pi = 3.14159265358979323846;
    const double* m = meas + (size_t)i * 9;
    double* f = filt + (size_t)i * 9;
    const double phi = m[3] / 180.0 * pi, theta = -m[4] / 180.0 * pi, psi = m[5] / 180.0 * pi;
    const double cph = cos(phi * 0.5), sph = sin(phi * 0.5);
    const double cth = cos(theta * 0.5), sth = sin(theta * 0.5);
    const double cps = cos(psi * 0.5), sps = sin(psi * 0.5);
    double qw = cph * cth * cps + sph * sth * sps;
    double qx = -(cps * cth * sph - sps * sth * cph);
    double qy = -(cps * sth * cph + sps * cth * sph);
    double qz = -(sps * cth * cph - cps * sth * sph);
    if (qw < 0) { qw = -qw; qx = -qx; qy = -qy; qz = -qz; }
    const double nrm = sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
    qw /= nrm; qx /= nrm; qy /= nrm; qz /= nrm;
    double ve[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double pk = m[a], pk1 = f[a], v1 = f[3 + a], v2 = f[6 + a];
        ve[a] = use_lpf ? (0.3306 * v1 - 0.02732 * v2 + 35.7 * pk - 35.7 * pk1) : (pk - pk1) / dt;
        f[a] = pk; f[6 + a] = v1; f[3 + a] = ve[a];
    }
    const double S11 = 2 * (qw * qw + qx * qx) - 1, S12 = 2 * (qx * qy + qw * qz), S13 = 2 * (qx * qz - qw * qy);
    const double S21 = 2 * (qx * qy - qw * qz), S22 = 2 * (qw * qw + qy * qy) - 1, S23 = 2 * (qy * qz + qw * qx);
    const double S31 = 2 * (qx * qz + qw * qy), S32 = 2 * (qy * qz - qw * qx), S33 = 2 * (qw * qw + qz * qz) - 1;
    double xc[13], uc[4], k1[13], k2[13], k3[13], k4[13], xt[13];
    xc[0] = m[0]; xc[1] = m[1]; xc[2] = m[2];
    xc[3] = qw; xc[4] = qx; xc[5] = qy; xc[6] = qz;
    xc[7] = S11 * ve[0] + S12 * ve[1] + S13 * ve[2];
    xc[8] = S21 * ve[0] + S22 * ve[1] + S23 * ve[2];
    xc[9] = S31 * ve[0] + S32 * ve[1] + S33 * ve[2];
    xc[10] = m[6]; xc[11] = m[7]; xc[12] = m[8];
#pragma unroll
    for (int e = 0; e < 13; e++) x_est[(size_t)i * 13 + e] = xc[e];
#pragma unroll
    for (int e = 0; e < 4; e++) uc[e] = u[(size_t)i * 4 + e];
    const double h = delay / steps;
    for (int s_ = 0; s_ < steps; s_++) {
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
    for (int e = 0; e < 13; e++) x_pred[(size_t)i * 13 + e] = xc[e];
}

// AoS [B][S][E] (external order) -> wave-blocked [wave][S][inst 0..3][E]; if perm13 the first 13
// entries of a row are permuted to the internal state order.  E == 4 fields are instance-major
// ([inst][S][4]) and handled by the same formula with a different block shape.
__device__ __forceinline__ size_t blk_index(int i, int s, int e, int S, int E, int v4b = 0) {
    if (E == 4 && !v4b) return ((size_t)i * S + s) * 4 + e;   // instance-major 4-vectors (Params.v4b = 0)
    return (((size_t)(i >> 2) * S + s) * 4 + (i & 3)) * E + e;
}
__global__ void k_put(int B, int S, int E, int perm13, const double* __restrict__ aos, double* __restrict__ blkp, int v4b) {
    const size_t n = (size_t)B * S * E;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(idx % E);
        const int s = (int)((idx / E) % S);
        const int i = (int)(idx / ((size_t)E * S));
        const int ei = (perm13 && e < 13) ? int_of(e) : e;
        blkp[blk_index(i, s, ei, S, E, v4b)] = aos[idx];
    }
}
__global__ void k_get(int B, int S, int E, int perm13, int s0, int Stot, const double* __restrict__ blkp,
                      double* __restrict__ aos, int v4b) {
    const size_t n = (size_t)B * S * E;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(idx % E);
        const int s = (int)((idx / E) % S);
        const int i = (int)(idx / ((size_t)E * S));
        const int ei = (perm13 && e < 13) ? int_of(e) : e;
        aos[idx] = blkp[blk_index(i, s0 + s, ei, Stot, E, v4b)];
    }
}
// Reference windows of the reference node generated on the device (acados_mpc.cpp:430-516), so
// that a tracking fleet needs no per-step host -> device reference traffic.
//   mode[i] 0 Regulation   : rows = [des_xyz(i), 1,0,0,0, 0 x 6, uss x 4]                 (:435-454)
//   mode[i] 1 Tracking     : rows k = 0..N <- traj[iter(i) + k] ; ++iter(i)                (:460-485)
//              (-> Position_Hold once iter >= n_rows - N; the window of that step is kept, :486)
//   mode[i] 2 Position_Hold: xyz of the last trajectory row, identity attitude, uss        (:494-513)
// One thread per (instance, stage); the policy update is done by stage-0 threads afterwards.
__global__ void k_windows(Params P, const double* __restrict__ traj, int n_rows, int* __restrict__ mode,
                          int* __restrict__ iter, const double* __restrict__ des, double uss) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int k = (int)(idx % (P.N + 1));
    const int i = (int)(idx / (P.N + 1));
    if (i >= P.B) return;
    // without a usable trajectory (n_rows < N + 1, traj may be NULL) Tracking / Position_Hold
    // instances are served as Regulation around des_xyz instead of dereferencing traj
    const int m = n_rows >= P.N + 1 ? mode[i] : 0;
    double row[17];
    bool write = true;
    if (m == 1) {
        const int it = iter[i];
        if (it < n_rows - P.N) {
            for (int e = 0; e < 17; e++) row[e] = traj[(size_t)(it + k) * 17 + e];
        } else {
            write = false;  // switches to Position_Hold; this step keeps the previous window
        }
    } else {
        const double* xyz = (m == 0) ? des + (size_t)i * 3 : traj + (size_t)(n_rows - 1) * 17;
        for (int e = 0; e < 17; e++) row[e] = 0.0;
        row[0] = xyz[0]; row[1] = xyz[1]; row[2] = xyz[2]; row[3] = 1.0;
        for (int e = 13; e < 17; e++) row[e] = uss;
    }
    if (!write) return;
    if (k < P.N) {
        double* y = P.yref + blk_index(i, k, 0, P.N, 17);
        for (int e = 0; e < 17; e++) y[e < 13 ? int_of(e) : e] = row[e];
    } else {
        double* y = P.yref_e + blk_index(i, 0, 0, 1, 13);
        for (int e = 0; e < 13; e++) y[int_of(e)] = row[e];
    }
}
__global__ void k_windows_advance(int B, int N, int n_rows, int* __restrict__ mode, int* __restrict__ iter) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B || mode[i] != 1 || n_rows < N + 1) return;
    if (iter[i] < n_rows - N) iter[i] += 1; else mode[i] = 2;
}

// Output stage of the reference node for a fleet (NMPC::iteration, acados_mpc.cpp:619-670), one
// vehicle per lane, straight from the iterate:
//   motvel [B][4] (int32)  = u0 truncated toward zero -- the int32 fields of PropellerSpeedsStamped
//                            (msg/PropellerSpeedsStamped.msg:2-5; assignment at acados_mpc.cpp:637-640)
//   cmd_vel [B][4]         = { pitch [deg] = +deg(theta), roll [deg] = -deg(phi), thrust [PWM] =
//                            (int)((mean(u1) * 1000 - 4070.3) / 0.2685), yaw rate [deg/s] = deg(x4.wz) }
//                            with (phi, theta) from the NORMALISED quaternion of x4 (:645-668; Euler
//                            formulas :384-404, PWM map :421-425, pi as defined at :106)
// u0 / u1 = inputs of stages 0 / 1, x4 = state of stage 4 (60 ms delay compensation, :624).
__global__ void k_postproc(Params P, double* __restrict__ cmd_vel, int* __restrict__ motvel) {
#pragma clang fp contract(off)   // the PWM value is truncated to int: keep the reference's rounding sequence
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.B) return;
    const double pi = 3.14159265358979323846;
    const double* u0 = P.uit + blk_index(i, 0, 0, P.N, 4, P.v4b);
    const double* u1 = P.uit + blk_index(i, 1, 0, P.N, 4, P.v4b);
    const double* x4 = P.xit + blk_index(i, 4, 0, P.N + 1, 13);
    double qw = x4[int_of(3)], qx = x4[int_of(4)], qy = x4[int_of(5)], qz = x4[int_of(6)];
    const double nrm = sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
    qw /= nrm; qx /= nrm; qy /= nrm; qz /= nrm;
    const double R31 = 2 * (qx * qz + qw * qy);
    const double R32 = 2 * (qy * qz - qw * qx);
    const double R33 = 2 * (qw * qw + qz * qz) - 1;
    const double phi = atan2(R32, R33), theta = -asin(R31);
    const double mean1 = (u1[0] + u1[1] + u1[2] + u1[3]) / 4;
    const int pwm = (int)(((mean1 * 1000) - 4070.3) / 0.2685);
    double* c = cmd_vel + (size_t)i * 4;
    c[0] = theta * 180.0 / pi;
    c[1] = -1.0 * (phi * 180.0 / pi);
    c[2] = (double)pwm;
    c[3] = x4[int_of(12)] * 180.0 / pi;
    if (motvel) {
        int* mv = motvel + (size_t)i * 4;
        for (int a = 0; a < 4; a++) mv[a] = (int)u0[a];
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
        for (int e = 0; e < 4; e++) P.uit[blk_index(i, k, e, P.N, 4, P.v4b)] = (mode == 1) ? hov : 0.0;
}

// cfnmpc_opts.reinit_failed: an instance whose last step ended in status 4 (factorisation not positive definite / not
// finite: its iterate has left the region where the Gauss-Newton QP is solvable, and re-linearising around the same iterate
// fails the same way step after step) restarts from x_k = its current x0, u_k = the input reference of its stage.
__global__ void k_reinit_failed(Params P) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.B || P.status[i] != 4) return;
    for (int k = 0; k <= P.N; k++)
        for (int e = 0; e < 13; e++) P.xit[blk_index(i, k, e, P.N + 1, 13)] = P.x0[blk_index(i, 0, e, 1, 13)];
    for (int k = 0; k < P.N; k++)
        for (int e = 0; e < 4; e++) P.uit[blk_index(i, k, e, P.N, 4, P.v4b)] = P.yref[blk_index(i, k, 13 + e, P.N, 17)];
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static inline int imin_h(int a, int b) { return a < b ? a : b; }
static inline int imax_h(int a, int b) { return a > b ? a : b; }
void launch_linearise(const Params& P, int chunks, hipStream_t st) {
    hipLaunchKernelGGL(k_linearise, dim3((P.NW + 15) / 16, chunks), dim3(64), 0, st, P);
}
#ifdef CFN_DEV
void launch_linearise_list(const Params& P, int chunks, hipStream_t st) {
    hipLaunchKernelGGL(k_linearise_list, dim3((P.NW + 15) / 16, chunks), dim3(64), 0, st, P);
}
#endif
void launch_linearise_clist(const Params& P, int chunks, int which, hipStream_t st) {
    hipLaunchKernelGGL(k_linearise_clist, dim3((P.NW + 15) / 16, chunks), dim3(64), 0, st, P, which);
}
// ev (optional, cfnmpc_set_profiling): events recorded after k_factor, after the forward sweep, after the compaction
void launch_factor_only(const Params& P, hipStream_t st);
void launch_qp_start(const Params& P, hipStream_t st, hipEvent_t* ev, bool skip_factor) {
    if (skip_factor) {}   // (sub-fleet pipeline: the backward sweep was launched on the start-solve stream)
    else if (P.fused && !P.lbs) launch_linfactor(P, st);   // fused start solve (cfnmpc_linfactor.hip); per-stage boxes: stored path
    else launch_factor_only(P, st);
    if (ev) (void)hipEventRecord(ev[0], st);
    if (P.lbs) {   // per-stage boxes: the row-group forward sweep carries them
        hipLaunchKernelGGL(k_forward_rg_sbox, dim3(P.NW), dim3(64), 0, st, P);
        hipLaunchKernelGGL(k_rank, dim3((P.B + 63) / 64), dim3(64), 0, st, P);
    } else if (P.forward_rg) {
        hipLaunchKernelGGL(k_forward_rg, dim3(P.NW), dim3(64), 0, st, P);
        hipLaunchKernelGGL(k_rank, dim3((P.B + 63) / 64), dim3(64), 0, st, P);
    } else {
#ifdef CFN_DEV
        if (P.forward_half) { hipLaunchKernelGGL(k_forward_half, dim3((P.B + 63) / 64), dim3(64), 0, st, P); } else
#endif
        if (P.fwd_split) hipLaunchKernelGGL(k_forward_p1, dim3((P.B + 63) / 64), dim3(64), 0, st, P);   // part two: launch_qp_ipm
        else hipLaunchKernelGGL(k_forward, dim3((P.B + 63) / 64), dim3(64), 0, st, P);
    }
    if (ev) (void)hipEventRecord(ev[1], st);
    hipLaunchKernelGGL(k_compact, dim3(N_BIN), dim3(256), 0, st, P);
    hipLaunchKernelGGL(k_scatter, dim3((P.B + 255) / 256), dim3(256), 0, st, P);
    if (ev) (void)hipEventRecord(ev[2], st);
}
#ifdef CFN_DEV
void launch_factor_chunk(const Params& P, hipStream_t st) {
    hipLaunchKernelGGL(k_factor_chunk, dim3(P.NW), dim3(64), 0, st, P);
}
#endif
void launch_factor_only(const Params& P, hipStream_t st) {
    // the start-solve kernels index the home 4-vectors in the wave-blocked layout unconditionally (cfnmpc_rg.hpp: load_stage,
    // factor_stage); solvers without it (partial condensing) have no path that leads here -- refuse rather than corrupt
    if (!P.v4b) { std::fprintf(stderr, "cfnmpc: k_factor needs the wave-blocked 4-vector layout (not a cond_N2 solver)\n"); return; }
    hipLaunchKernelGGL(k_factor, dim3(P.NW), dim3(64), 0, st, P);
}
void launch_cforward(const Params& P, hipStream_t st) {
    hipLaunchKernelGGL(k_cforward, dim3((P.B + 63) / 64), dim3(64), 0, st, P);
}
// ev (optional): event recorded after the active-set kernels (before the interior-point launch for what they left)
void launch_qp_ipm(const Params& P, hipStream_t st, hipEvent_t* ev) {
    if (P.fused == 1 && !P.lbs) {
        // fused start solve: no stored stage blocks -- the constrained instances are re-linearised into their compact
        // store first (and the interior-point fall-back rows once more after k_ipm_list has moved them to new slots)
        launch_linearise_clist(P, P.clist_chunks, 0, st);
        if (P.active_set) {
            hipLaunchKernelGGL(k_as_cst, dim3(P.NW), dim3(64), 0, st, P);   // (lists its fall-back rows itself)
            if (ev) (void)hipEventRecord(ev[0], st);
            if (P.ipm_listed) launch_linearise_clist(P, P.clist_chunks, 1, st);
            hipLaunchKernelGGL(k_ipm_rest_cst, dim3(imax_h(1, imin_h(P.NW, P.as_grid / 2))), dim3(64), 0, st, P);
        } else {
            if (ev) (void)hipEventRecord(ev[0], st);
            hipLaunchKernelGGL(k_ipm_cst, dim3(P.NW), dim3(64), 0, st, P);
        }
        return;
    }
    if (P.lbs) {   // per-stage boxes: monolithic kernels instantiated for them
        if (P.active_set) {
            hipLaunchKernelGGL(k_as_sbox, dim3(P.NW), dim3(64), 0, st, P);   // (lists its fall-back rows itself)
            if (ev) (void)hipEventRecord(ev[0], st);
            hipLaunchKernelGGL(k_ipm_rest_sbox, dim3(imax_h(1, imin_h(P.NW, P.as_grid / 2))), dim3(64), 0, st, P);
        } else {
            if (ev) (void)hipEventRecord(ev[0], st);
            hipLaunchKernelGGL(k_ipm_sbox, dim3(P.NW), dim3(64), 0, st, P);
        }
    } else if (P.active_set && P.as_passes != 0) {
        // as_passes > 0: level-synchronous active-set passes -- pairs (factor, forward) of one solve each, one
        // launch loops over the remaining solves in-wave; as_passes < 0: every solve in one launch.  Then commit,
        // retries over a longer head, interior point for the rest.
        const int G = imax_h(1, imin_h(P.as_grid, P.NW));
        if (P.as_passes == -2 && P.as_dense) {
            // heads of at most 16 stages: head-condensed dense solves (one row per wavefront, one wavefront per SIMD) on the
            // caller's stream; the rows with longer heads (the first P.nipm[NI_LONG16] of the list: none, or a handful) keep the Riccati
            // form of the iteration in k_as_solves, forked onto the solver's side stream -- two latency chains side by side.
            // Split forward sweep: its second part (stages [24, N) of EVERY instance) runs on a second side stream beside both;
            // rows with heads of 24 stages read nothing behind stage 24 and stay beside it, the rows with longer heads (rare) are
            // left to the retry kernel.
            hipStream_t side = (hipStream_t)P.as_side, side2 = (hipStream_t)P.as_side2;
            // (a step being CAPTURED into a graph -- cfnmpc_opts.step_graph -- keeps everything on the capture stream: forked streams
            //  inside a re-captured graph whose predecessor is still in flight crashed the runtime in one run of four)
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) side = side2 = nullptr;
            const bool p1_ran = P.fwd_split && !P.lbs && !P.forward_rg;     // (launch_qp_start: the sweep's first part only)
            const bool split = p1_ran && side && side2;
            if (side) (void)hipEventRecord((hipEvent_t)P.as_fork, st);
            if (split) {
                Params PA = P;
                PA.as_range = 1;
                (void)hipStreamWaitEvent(side2, (hipEvent_t)P.as_fork, 0);
                hipLaunchKernelGGL(k_forward_p2, dim3((P.B + 63) / 64), dim3(64), 0, side2, P);
                // (the rows with heads behind the split point -- PB's range: rare -- are left to k_as_retry below: a launch behind
                //  part two, which is the longest chain of this group, cost 5 + 8 us per step whether it had work or not)
                (void)hipEventRecord((hipEvent_t)P.as_join2, side2);
                (void)hipStreamWaitEvent(side, (hipEvent_t)P.as_fork, 0);
                hipLaunchKernelGGL(k_as_solves, dim3(P.NW), dim3(64), 0, side, PA);
                (void)hipEventRecord((hipEvent_t)P.as_join, side);
            } else {
                if (p1_ran) hipLaunchKernelGGL(k_forward_p2, dim3((P.B + 63) / 64), dim3(64), 0, st, P);   // no side streams: in order
                if (side && !p1_ran) {
                    (void)hipStreamWaitEvent(side, (hipEvent_t)P.as_fork, 0);
                    hipLaunchKernelGGL(k_as_solves, dim3(P.NW), dim3(64), 0, side, P);
                    (void)hipEventRecord((hipEvent_t)P.as_join, side);
                } else {
                    side = nullptr;
                    Params PA = P;
                    PA.as_range = p1_ran ? 1 : 0;   // (behind a split sweep the same rows as with the side streams: bit-identical steps)
                    hipLaunchKernelGGL(k_as_solves, dim3(P.NW), dim3(64), 0, st, PA);
                }
            }
            launch_as_dense(P, imax_h(1, imin_h(P.as_grid / 2, P.NW * 4)), st);
            if (side) (void)hipStreamWaitEvent(st, (hipEvent_t)P.as_join, 0);
            if (split) (void)hipStreamWaitEvent(st, (hipEvent_t)P.as_join2, 0);
        } else if (P.as_passes == -2) {
            hipLaunchKernelGGL(k_as_solves, dim3(P.NW), dim3(64), 0, st, P);
        }
#ifdef CFN_DEV
        else if (P.as_passes < 0) {
            hipLaunchKernelGGL(k_asp_all, dim3(imax_h(1, imin_h(P.as_grid / 2, P.NW))), dim3(64), 0, st, P);
        } else {
            hipLaunchKernelGGL(k_asf_first, dim3(G), dim3(64), 0, st, P);
            hipLaunchKernelGGL(k_asw_first, dim3(G), dim3(64), 0, st, P);
            for (int p = 1; p < P.as_passes; p++) {
                hipLaunchKernelGGL(k_asf, dim3(G), dim3(64), 0, st, P, p);
                hipLaunchKernelGGL(k_asw, dim3(G), dim3(64), 0, st, P, p);
            }
            if (P.as_passes < AS_MAX_SOLVES)
                hipLaunchKernelGGL(k_asp, dim3(imax_h(1, G / 2)), dim3(64), 0, st, P, P.as_passes, AS_MAX_SOLVES - P.as_passes);
        }
#endif
        if (P.NW <= 2 * P.as_grid) hipLaunchKernelGGL(k_ascommit1, dim3(imax_h(1, imin_h(P.as_grid / 2, P.NW))), dim3(64), 0, st, P);
        else hipLaunchKernelGGL(k_ascommit, dim3(G), dim3(64), 0, st, P);
        // (late rows of a split forward sweep: k_as_retry and k_ipm_rest count them in, P.nipm[0] + P.nipm[NI_LATE].  Tried: both modes in
        //  ONE launch for small fleets, where each normally finds nothing to do and an empty launch costs 5 - 7 us -- either half
        //  alone runs, the combined kernel aborts on the device; not pursued)
        hipLaunchKernelGGL(k_as_retry, dim3(P.NW), dim3(64), 0, st, P);
        if (ev) (void)hipEventRecord(ev[0], st);
        if (P.ipm_listed) hipLaunchKernelGGL(k_ipm_list, dim3(1), dim3(1024), 0, st, P);
        hipLaunchKernelGGL(k_ipm_rest, dim3(imax_h(1, imin_h(P.NW, P.as_grid / 2))), dim3(64), 0, st, P);
    } else if (P.active_set) {
        hipLaunchKernelGGL(k_as, dim3(P.NW), dim3(64), 0, st, P);   // (lists its fall-back rows itself)
        if (ev) (void)hipEventRecord(ev[0], st);
        hipLaunchKernelGGL(k_ipm_rest, dim3(imax_h(1, imin_h(P.NW, P.as_grid / 2))), dim3(64), 0, st, P);
    } else {
        if (ev) (void)hipEventRecord(ev[0], st);
        hipLaunchKernelGGL(k_ipm, dim3(P.NW), dim3(64), 0, st, P);
    }
}
void launch_qp(const Params& P, hipStream_t st, hipEvent_t* ev) {
    launch_qp_start(P, st, ev);
    launch_qp_ipm(P, st, ev ? ev + 3 : nullptr);
}
void launch_sim(int B, const double* x, const double* u, double T, int steps, double* xn, hipStream_t st) {
    hipLaunchKernelGGL(k_sim, dim3((B + 255) / 256), dim3(256), 0, st, B, x, u, T, steps, xn);
}
void launch_estimate(int B, const double* meas, double* filt, const double* u, double dt, int use_lpf, double delay,
                     int steps, double* x_est, double* x_pred, hipStream_t st) {
    hipLaunchKernelGGL(k_estimate, dim3((B + 255) / 256), dim3(256), 0, st, B, meas, filt, u, dt, use_lpf, delay, steps,
                       x_est, x_pred);
}
static inline int grid_for(size_t n) {
    size_t g = (n + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}
void launch_put(int B, int S, int E, int perm13, const double* aos, double* blkp, hipStream_t st, int v4b) {
    hipLaunchKernelGGL(k_put, dim3(grid_for((size_t)B * S * E)), dim3(256), 0, st, B, S, E, perm13, aos, blkp, v4b);
}
void launch_get(int B, int S, int E, int perm13, int s0, int Stot, const double* blkp, double* aos, hipStream_t st, int v4b) {
    hipLaunchKernelGGL(k_get, dim3(grid_for((size_t)B * S * E)), dim3(256), 0, st, B, S, E, perm13, s0, Stot, blkp, aos, v4b);
}
void launch_windows(const Params& P, const double* traj, int n_rows, int* mode, int* iter, const double* des,
                    double uss, hipStream_t st) {
    const size_t n = (size_t)P.B * (P.N + 1);
    hipLaunchKernelGGL(k_windows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P, traj, n_rows, mode, iter, des, uss);
    hipLaunchKernelGGL(k_windows_advance, dim3((P.B + 255) / 256), dim3(256), 0, st, P.B, P.N, n_rows, mode, iter);
}
void launch_postproc(const Params& P, double* cmd_vel, int* motvel, hipStream_t st) {
    hipLaunchKernelGGL(k_postproc, dim3((P.B + 255) / 256), dim3(256), 0, st, P, cmd_vel, motvel);
}
void launch_reinit_failed(const Params& P, hipStream_t st) {
    hipLaunchKernelGGL(k_reinit_failed, dim3((P.B + 255) / 256), dim3(256), 0, st, P);
}
void launch_init_iterate(const Params& P, int mode, hipStream_t st) {
    hipLaunchKernelGGL(k_init_iterate, dim3((P.B + 255) / 256), dim3(256), 0, st, P, mode);
}

}  // namespace cfn
