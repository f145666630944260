// cfnmpc_asdense.hip -- active-set solves of the constrained instances on the HEAD-CONDENSED dense QP (gfx950, FP64).
//
// Where it sits (cfnmpc_kernels.hip, launch_qp_ipm): the constrained rows of an RTI step are solved on a head of 4 .. 32 or all
// N stages (active horizon: the unconstrained tail keeps the start solve's feedback law and enters through its cost-to-go
// checkpoint P_head).  The Riccati form of a primal-dual active-set solve (qp_wave: one factorisation + one forward sweep over
// the head per solve) is a chain of 2 x head dependent stages of ~1 - 1.6 us each; on the bench workload 99 % of the constrained
// rows have heads of 8 - 16 stages, i.e. 32 - 64 inputs, and need 1 - 5 solves: 30 - 220 us per row, which IS the duration of the
// active-set kernel wherever rows are fewer than SIMDs, and 17 % of the step at 65 536 instances.  For heads of at most 16 stages
// this kernel solves the SAME QP, by the SAME active-set iteration, in dense form -- the reference's own solver plan condenses
// too (PARTIAL_CONDENSING_HPIPM, generate_c_code.py:140) --:
//
//   1. H = R + Gamma' Q Gamma of the head in delta form (dx_0 = 0, stage cost Q on dx_1 .. dx_{h-1}, P_head on dx_h), n = 4 h
//      inputs, ONE COLUMN PER LANE: lane j = input (k_j, a_j) propagates x_{k+1} = A_k x_k (+ B_k[:, a_j] at k = k_j) forward and
//      the adjoint lam_i = Q x_i + A_i' lam_{i+1}, H[(i, .), j] = B_i' lam_{i+1} (+ R at its own entry) backward -- 259 fused
//      multiply-adds per stage and lane, the stage matrices being DPP broadcast sources (cfnmpc_dense_dpp.hpp) straight from the
//      home blocks' row-distributed form.  The backward sweep needs the forward states: eight of them are kept in registers and
//      heads of more than eight stages are done in two halves (the first half's states recomputed).
//   2. + 3. the primal-dual active-set iteration on the SWEPT matrix: every lane holds its row of H in registers; Goodnight's
//      symmetric sweep on the free inputs F leaves -H_FF^-1, H_FF^-1 H_FA and the Schur complement of the fixed inputs in
//      place, so one solve with the inputs A fixed at c = bound - v0 is ONE matrix-vector product (delta_F = -W[F, A] c_A,
//      multipliers = W[A, A] c_A), and since sweeps commute and are reversible the matrix follows the active set: the first
//      iteration sweeps the initially free inputs, every later one only the inputs whose class changed.  Same classification
//      rule, same sequence of sets and same solve counts as qp_wave and the CPU restatements of the test suite; a stationary
//      set is the KKT point of the strictly convex QP.
//   4. dx of the head by one forward sweep; du / dx / settled flag / solve count handed to k_ascommit exactly as k_as_solves
//      does (roll-out, tail verification, retries and the interior-point fall-back are unchanged).
//
// One wavefront = one row; rows with longer heads stay with k_as_solves (the compaction lists them first: P.nipm[NI_LONG16] rows).
// LDS: 32 KB (staged stage matrices, then H) + 2 KB per wavefront -> four per compute unit.  Scalar input box only.
#include <hip/hip_runtime.h>

#include "cfnmpc_rg.hpp"
#include "cfnmpc_dense_dpp.hpp"

namespace cfn {

namespace {

constexpr int DN = 64;   // columns of the dense store (a head of 16 stages)

struct DenseLds {
    double G[DN * DN];      // the staged stage matrices, then H: element (r, j) at r * DN + j (lane j: conflict-free)
    double cv[DN];          // delta of the settled row (read by the dx sweep)
    double pm[16 * 13];     // cost-to-go at the head, lane-distributed: pm[L * 13 + c] = P_head[L][c]
};
// During the build the H block of stage k (rows 4 k .. 4 k + 3: 256 doubles) first holds that stage's matrices in the lane-distributed
// form of the home blocks -- lane L of a DPP row: A[L][3 .. 12] | B[L][0 .. 3] at k * 256 + L * 14 -- staged once from HBM with all loads
// in flight together; the backward visit of stage k is the last reader of that block and overwrites it with its rows of H.
constexpr int ST_BLK = 256, ST_ROW = 14;
static_assert(16 * ST_ROW <= ST_BLK && ST_BLK == 4 * DN, "staged stage matrices fit the stage's H rows");

struct StageOps { double ac[10], br[4]; };
struct BuildVisit { int kind, stage, slot; };   // 0: forward, 1: forward with the new state kept in hist[slot], 2: backward
template <bool TWO>
__host__ __device__ constexpr BuildVisit build_visit(int v) {
    if (TWO) {
        if (v < 8) return {0, v, 0};
        if (v < 16) return {1, v, v - 8};
        if (v < 24) return {2, 15 - (v - 16), 0};
        if (v < 32) return {1, v - 24, v - 24};
        return {2, 7 - (v - 32), 0};
    }
    if (v < 8) return {1, v, v};
    return {2, 7 - (v - 8), 0};
}
constexpr int AS_MAX_SOLVES_DENSE = 12;   // the iteration cap of qp_wave (AS_MAX_SOLVES)

__device__ __forceinline__ void load_ops(const Params& P, const Lane& t, const int k, StageOps& o) {
    ld_ar(blkab(P, P.AR, t, k, SZ_A), t, o.ac);
    ld_rows4(blkab(P, P.BR, t, k, SZ_B), t, o.br);
}
__device__ __forceinline__ void lds_ops(const double* G, const Lane& t, const int k, StageOps& o) {
    const double* b = G + k * ST_BLK + t.L * ST_ROW;
    SFOR(g, 0, 10, { o.ac[g] = b[g]; });
    SFOR(a, 0, 4, { o.br[a] = b[10 + a]; });
}
// stage matrices of the head and the cost-to-go at the head -> LDS (one exposed HBM latency per row)
__device__ __forceinline__ void dense_stage(const Params& P, const Lane& t, const int head, const int chk, double* G, double* pm) {
    StageOps o[4];
    SFOR(b, 0, 4, { load_ops(P, t, imin(4 * b + t.row, head - 1), o[b]); });   // DPP row r of batch b: stage 4 b + r
    double pr[13];
    if (chk >= 0) {
        const gdouble* pc = gm(P.Pchk) + ((size_t)t.wave * N_CHK + chk) * SZ_PP;
        SFOR(c, 0, 13, {
            const double v = pc[pchk_at(c, t.q, imin(t.L, 12))];
            pr[c] = t.L < 13 ? v : 0.0;
        });
    } else {
        SFOR(c, 0, 13, { pr[c] = (t.L == c) ? P.WN[ext_of(c)] : 0.0; });
    }
    SFOR(b, 0, 4, {
        const int k = 4 * b + t.row;
        if (k < head) {
            double* d = G + k * ST_BLK + t.L * ST_ROW;
            SFOR(g, 0, 10, { d[g] = o[b].ac[g]; });
            SFOR(a, 0, 4, { d[10 + a] = o[b].br[a]; });
        }
    });
    if (t.row == 0) SFOR(c, 0, 13, { pm[t.L * 13 + c] = pr[c]; });
}

// ---- 1. H column of this lane -> S.G[(i * 4 + a) * DN + lane] ------------------------------------------------------------------
// TWO: head > 8 (two halves).  All lanes run every stage (lockstep); a lane's state is zero up to its own stage.
template <bool TWO>
__device__ __forceinline__ void dense_build(const Params& P, const Lane& t, const int head, const int chk, DenseLds& S) {
    const int lane = threadIdx.x;
    const int kj = lane >> 2, aj = lane & 3;
    double Qi[13];
    SFOR(j, 0, 13, { Qi[j] = P.W[ext_of(j)]; });
    const double Rj = aj == 0 ? P.W[13] : (aj == 1 ? P.W[14] : (aj == 2 ? P.W[15] : P.W[16]));
    // this lane's own column of B (injected at its stage)
    double binj[13];
    SFOR(i, 0, 13, { binj[i] = S.G[imin(kj, head - 1) * ST_BLK + i * ST_ROW + 10 + aj]; });
    double pr[13];   // pr[c]@lane(r) = P_head[r][c] (read from the staged copy one visit before its use)
    auto load_pr = [&]() { SFOR(c, 0, 13, { pr[c] = S.pm[t.L * 13 + c]; }); };
    auto fwd = [&](const int k, double (&x)[13], const StageOps& o) {
        double xn[13];
        SFOR(i, 0, 3, { xn[i] = x[i]; });
        SFOR(i, 3, 13, { xn[i] = 0.0; });
        dense_fwd(xn, x, o.ac);
        const double inj = (k == kj) ? 1.0 : 0.0;
        SFOR(i, 0, 13, { x[i] = __builtin_fma(inj, binj[i], xn[i]); });
    };
    auto bwd = [&](const int i, double (&lam)[13], const double (&xi)[13], const StageOps& o) {
        double h[4] = {0.0, 0.0, 0.0, 0.0};
        dense_bwd_h(h, lam, o.br);
        SFOR(a, 0, 4, { S.G[(i * 4 + a) * DN + lane] = h[a] + ((i == kj && a == aj) ? Rj : 0.0); });
        double ln[13];
        SFOR(c, 0, 3, { ln[c] = __builtin_fma(Qi[c], xi[c], lam[c]); });
        SFOR(c, 3, 13, { ln[c] = Qi[c] * xi[c]; });
        dense_bwd_a(ln, lam, o.ac);
        SFOR(c, 0, 13, { lam[c] = ln[c]; });
    };
    // kept states: a backward stage i reads x_i, i.e. the LAST state of each forward pass is never read back -- seven slots per
    // pass, the eighth holds the hand-over state x_8 of the two-half build
    double x[13], lam[13], hist[8][13], zero[13];
    SFOR(i, 0, 13, { x[i] = 0.0; lam[i] = 0.0; zero[i] = 0.0; });
    // The visits of the stage matrices, in order (compile-time list; the operands of visit v + 1 are in flight while visit v
    // is computed: two stage buffers used alternately):
    //   TWO : F 0..7 | F 8..15 (x_9 .. x_16 kept) | B 15..8 | F 0..7 (x_1 .. x_8 kept) | B 7..0
    //   else: F 0..7 (kept) | B 7..0
    // visits of stages >= head are skipped (their loads, clamped, are harmless)
    constexpr int NV = TWO ? 40 : 16, FIRST_B = TWO ? 16 : 8;
    StageOps o0, o1;
    lds_ops(S.G, t, 0, o0);
    SFOR(v, 0, NV, {
        StageOps& cur = (v & 1) ? o1 : o0;
        StageOps& nxt = (v & 1) ? o0 : o1;
        // (the visits are straight-line code: without a fence the compiler forms the addresses of all forty visits up front and
        //  hoists their loads as far as it can -- 400 spilled registers; the opaque stage index pins each visit's loads to its place)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (v + 1 < NV) {
            int kn = imin(build_visit<TWO>(v + 1).stage, head - 1);
            asm volatile("" : "+s"(kn));
            lds_ops(S.G, t, kn, nxt);
        }
        constexpr BuildVisit vv = build_visit<TWO>(v);
        if constexpr (v == FIRST_B - 1) load_pr();
        if constexpr (v == FIRST_B) dense_px(lam, x, pr);            // lam_head = P_head x_head
        if constexpr (TWO && v == 24) { SFOR(i, 0, 13, { x[i] = 0.0; }); }
        if constexpr (vv.kind != 2) {
            if (vv.stage < head) fwd(vv.stage, x, cur);
            if constexpr (vv.kind == 1 && vv.slot < 7) { SFOR(i, 0, 13, { hist[vv.slot][i] = x[i]; }); }
            if constexpr (TWO && v == 7) { SFOR(i, 0, 13, { hist[7][i] = x[i]; }); }
        } else {
            constexpr int i = vv.stage;
            if (i < head) {
                if constexpr (TWO && v < 24) {          // first half: x_8 from the hand-over copy, x_9 .. x_15 from the kept states
                    if constexpr (i == 8) bwd(i, lam, hist[7], cur);
                    else bwd(i, lam, hist[i - 9], cur);
                } else {
                    if constexpr (i == 0) bwd(i, lam, zero, cur);
                    else bwd(i, lam, hist[i - 1], cur);
                }
            }
        }
    });
}

// ---- 2. + 3.: the active-set iteration on the SWEPT matrix; NB = sixteens of inputs (n = 16 NB >= 4 head) -------------------
// Goodnight's symmetric sweep on an index set F turns H = [H_FF H_FA; H_AF H_AA] into
//     [ -H_FF^-1 ,  H_FF^-1 H_FA ;  H_AF H_FF^-1 ,  H_AA - H_AF H_FF^-1 H_FA ],
// i.e. exactly what one active-set solve with the inputs A fixed at c needs: with val = W[:, A] c_A the free inputs move by
// delta_F = -val_F and the multipliers of the fixed ones are val_A (Schur complement).  Sweeps commute and are reversible (the reverse
// sweep differs in one sign), so the matrix FOLLOWS the active set: the first iteration sweeps the initially free inputs, every
// later one only the inputs that changed their class -- a few rank-one updates instead of a new factorisation -- and each solve is
// one matrix-vector product.  A sweep: every lane holds its row; the pivot column is the lanes' own element k (the swept matrix
// stays symmetric), which reaches the 16-lane DPP rows by one cross-lane permute per sixteen columns; its pivot by v_readlane.
// returns the number of solves (> 0: settled, delta in S.cv), 0: not settled / a pivot of the wrong sign
template <int NB>
__device__ __forceinline__ int dense_solve(const Params& P, const int head, const double v0, const double uk, DenseLds& S) {
    constexpr int n = 16 * NB;
    const int lane = threadIdx.x;
    const int L = lane & 15;
    const int nr = 4 * head;                 // real inputs; lanes / columns behind them are padding (identity, never swept)
    double w[n];
    SFOR(c, 0, n, { w[c] = S.G[c * DN + lane]; });
    SFOR(c, 0, n, { if (lane >= nr || c >= nr) w[c] = (lane == c) ? 1.0 : 0.0; });
    // element state of this lane's input
    const bool real = lane < nr;
    const double lb = real ? P.u_min - uk : -1e300, ub = real ? P.u_max - uk : 1e300;
    int cls = v0 < lb ? 1 : (v0 > ub ? 2 : 0);
    double c = cls == 1 ? lb - v0 : (cls == 2 ? ub - v0 : 0.0);
    double delta = 0.0;
    const unsigned long long realmask = nr >= 64 ? ~0ull : ((1ull << nr) - 1ull);
    unsigned long long swept = 0ull;
    bool ok = true;
    int solves = 0;
    bool done = false;
    for (int it = 1; it <= AS_MAX_SOLVES_DENSE && !done; it++) {
        solves = it;
        const bool act = cls != 0;
        const unsigned long long free_now = realmask & ~__ballot(act);
        const unsigned long long chg = swept ^ free_now;
        // (straight-line walk over the n unrolled sweep bodies with a wave-uniform skip each.  Measured alternatives, both slower by
        //  ~14 us per row because the register allocator then spills the rows: a loop over the due indices with a binary dispatch
        //  to the body, and prefetching the next pivot column inside the previous sweep)
        SFOR(k, 0, n, {
            if ((chg >> k) & 1ull) {
                const bool rev = !((free_now >> k) & 1ull);          // the input became active: take it out of the swept set
                const double col = w[k];
                double creg[NB];
                SFOR(tt, 0, NB, { creg[tt] = __shfl(col, L + 16 * tt); });
                const double d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(col), k), __builtin_amdgcn_readlane(__double2loint(col), k));
                ok = ok && (rev ? (d < 0.0) : (d > 0.0));
                const double rinv = rcp_nr(d);
                const double tk = col * rinv;
                const double ti = (lane == k) ? (rev ? 1.0 + rinv : 1.0 - rinv) : tk;   // lane k: its own row ends as +- row / d
                const double negt = -ti;
                SFOR(tt, 0, NB, { rank1bc16(&w[16 * tt], negt, creg[tt]); });
                w[k] = (lane == k) ? -rinv : (rev ? -tk : tk);
            }
        });
        swept = free_now;
        if (!ok) return 0;
        // val = W[:, A] c_A
        const double cc = act ? c : 0.0;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        SFOR(tt, 0, NB, {
            const double cr = __shfl(cc, L + 16 * tt);
            dot16bc4(acc, &w[16 * tt], cr);
        });
        const double val = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        const double dl = act ? c : -val;
        delta = dl;
        const double vn = v0 + dl;
        int nc;
        if (cls == 0) nc = vn < lb ? 1 : (vn > ub ? 2 : 0);
        else if (cls == 1) nc = val > 0.0 ? 1 : 0;       // multiplier of a fixed input: stays while it pushes outward
        else nc = val < 0.0 ? 2 : 0;
        const bool changed = __any(real && nc != cls);
        cls = real ? nc : 0;
        c = cls == 1 ? lb - v0 : (cls == 2 ? ub - v0 : 0.0);
        done = !changed;
    }
    __syncthreads();
    S.cv[lane] = delta;
    __syncthreads();
    return done ? solves : 0;
}

}  // namespace

#ifdef CFN_PROF
// (development builds with -DCFN_PROF: phases of the longest row and sums over all rows, wall-clock ticks of 10 ns)
__device__ unsigned long long g_dprof[32];
#define DPROF(i) { const unsigned long long now_ = wall_clock64(); dacc[i] += now_ - dlast; dlast = now_; }
#else
#define DPROF(i)
#endif

KALIGN __global__ __launch_bounds__(64) void k_as_dense(Params P) {
    __shared__ DenseLds S;
    const int lane = threadIdx.x;
    const int N = P.N;
    const int nipm = gm(P.nipm)[0], nbig = gm(P.nipm)[NI_LONG16];
    for (int slot = nbig + (int)blockIdx.x; slot < nipm; slot += (int)gridDim.x) {
#ifdef CFN_PROF
        unsigned long long dacc[6] = {0, 0, 0, 0, 0, 0}, dlast = wall_clock64();
        const unsigned long long dstart = dlast;
#endif
        const int inst = gm(P.ilist)[slot];
        const int head = gm(P.head)[inst];           // <= 16 (or N <= 16): guaranteed by the list order (k_scatter)
        const double viol = gm(P.viol)[inst];
        // rows far outside the box skip the active-set iteration (as qp_wave): straight to the interior point
        const bool try_as = viol > 0.0 && !(P.as_skip_viol > 0.0 && viol > P.as_skip_viol * (P.u_max - P.u_min));
        if (!try_as || head < 1 || head > 16) {
            if (lane == 0) gm(P.asst)[slot] = 0;
            continue;
        }
        int chk = -1;
        SFOR(c, 0, N_CHK, { if (head == chk_stage(c) && head < N) chk = c; });
        // every DPP row of the wave addresses the same instance (the stage matrices are broadcast sources inside a row)
        const Lane t = lane_indirect(P, inst, true);
        const Lane& t_ = t;
        // this lane's input (k_j, a_j): unconstrained minimiser and iterate, in flight during the build
        const size_t e4 = i4(P, t, imin(lane >> 2, head - 1), lane & 3);
        const double v0 = lane < 4 * head ? gm(P.v)[e4] : 0.0;
        const double uk = lane < 4 * head ? gm(P.uit)[e4] : 0.0;
        __syncthreads();
        dense_stage(P, lane_opaque(t), head, chk, S.G, S.pm);
        __syncthreads();
        DPROF(0)
        if (head > 8) dense_build<true>(P, lane_opaque(t), head, chk, S);
        else dense_build<false>(P, lane_opaque(t), head, chk, S);
        __syncthreads();
        DPROF(1)
        int solves;
        if (head <= 4) solves = dense_solve<1>(P, head, v0, uk, S);
        else if (head <= 8) solves = dense_solve<2>(P, head, v0, uk, S);
        else if (head <= 12) solves = dense_solve<3>(P, head, v0, uk, S);
        else solves = dense_solve<4>(P, head, v0, uk, S);
        DPROF(2)
        if (solves > 0) {
            // du of the head -> the compact slot's P.dva; dx_1 .. dx_head -> P.czdx (what k_ascommit reads)
            const double dl = S.cv[lane];
            if (lane < 4 * head) gm(P.dva)[(size_t)slot * N * 4 + lane] = dl;
            // the stage matrices once more (the H store is free again): staged with all loads in flight, then the sequential sweep
            __syncthreads();
            const Lane t = lane_opaque(t_);   // (the sweep's addresses stay inside it)
            {
                StageOps o[4];
                SFOR(b, 0, 4, { load_ops(P, t, imin(4 * b + t.row, head - 1), o[b]); });
                SFOR(b, 0, 4, {
                    const int k = 4 * b + t.row;
                    if (k < head) {
                        double* d = S.G + k * ST_BLK + t.L * ST_ROW;
                        SFOR(g, 0, 10, { d[g] = o[b].ac[g]; });
                        SFOR(a, 0, 4, { d[10 + a] = o[b].br[a]; });
                    }
                });
            }
            __syncthreads();
            double x = 0.0;
            StageOps o0, o1;
            lds_ops(S.G, t, 0, o0);
            auto step = [&](const StageOps& o, const int k) {
                double xn = t.L < 3 ? x : 0.0;
                dotbc<10, 3>(xn, o.ac, x);
                SFOR(a, 0, 4, { xn = __builtin_fma(o.br[a], S.cv[k * 4 + a], xn); });
                x = xn;
                if (t.row == 0 && t.L < 13) gm(P.czdx)[((size_t)slot * (N + 1) + k + 1) * 13 + t.L] = x;
            };
            for (int k = 0; k < head; k += 2) {
                lds_ops(S.G, t, imin(k + 1, head - 1), o1);
                step(o0, k);
                if (k + 1 >= head) break;
                lds_ops(S.G, t, imin(k + 2, head - 1), o0);
                step(o1, k + 1);
            }
        }
        if (lane == 0) {
            gm(P.asst)[slot] = solves > 0 ? 1 : 0;
            if (solves > 0) gm(P.iters)[inst] = solves;
        }
#ifdef CFN_PROF
        DPROF(3)
        if (lane == 0) {
            const unsigned long long tot = dlast - dstart;
            const unsigned long long old = atomicMax(&g_dprof[8], tot);
            if (tot > old) { for (int i = 0; i < 6; i++) g_dprof[i] = dacc[i]; g_dprof[11] = solves; g_dprof[12] = head; }
            atomicAdd(&g_dprof[9], tot); atomicAdd(&g_dprof[10], 1ull); atomicAdd(&g_dprof[13], (unsigned long long)solves);
            for (int i = 0; i < 6; i++) atomicAdd(&g_dprof[16 + i], dacc[i]);
        }
#endif
    }
}

#ifdef CFN_PROF
void debug_dprof_read(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dprof), sizeof(unsigned long long) * 32);
    if (reset) { unsigned long long z[32] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dprof), z, sizeof z); }
}
#endif
void launch_as_dense(const Params& P, int grid, hipStream_t st) {
    hipLaunchKernelGGL(k_as_dense, dim3(grid), dim3(64), 0, st, P);
}

}  // namespace cfn
