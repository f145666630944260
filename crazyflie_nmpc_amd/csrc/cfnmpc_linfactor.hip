// cfnmpc_linfactor.hip -- FUSED start solve, backward: Gauss-Newton linearisation INSIDE the Riccati factorisation
// (gfx950, FP64).  The stage blocks (A, B, b) never touch HBM.
//
// k_linearise writes 162 doubles per instance and stage that k_factor reads back one launch later -- 8.5 of the 17.8 GB a
// 65 536-instance RTI step moves.  Here the factorisation's own wavefront (row groups: one NMPC
// instance per 16-lane DPP row, four per wave, cfnmpc_ws.hpp) produces them stage by stage on its way backward:
//
//   * every lane of a row integrates the nominal RK4 step of its instance (redundantly: the 16 lanes share one
//     instruction stream) and ONE column of the forward sensitivities through the four RK points -- lanes 0..9 the
//     state columns v | q | w (internal columns 3..12; the position columns are the identity and are not stored),
//     lanes 10..13 the four input columns -- with ONE Jacobian point live at a time (the role of acados sim_erk + CasADi
//     forw_vde, acados_mpc.cpp:84; model: export_ode_model.py:85-97 as restated in cfnmpc_model.hpp);
//   * the 14 columns (lane c holds column c) go through the LDS tile that also transposes W = P A and come back in the
//     row form factor_stage() consumes (lane i holds row i of A and of B); Phi travels the same way, b = Phi - x_{k+1};
//   * factor_stage() -- unchanged, cfnmpc_rg.hpp -- does the augmented Riccati stage and stores gain + feed-forward;
//     the cost-to-go checkpoints of the active-horizon QP are written as k_factor writes them.
//
// Per instance and stage the kernel READS x_k, u_k, yref_k (34 doubles) and WRITES K, d (56 doubles): ~0.75 KB against
// k_linearise + k_factor's 2.9 KB.  What it pays: the column types' sparsity (all lanes run the general column) and a
// 16-fold redundant nominal step -- ~1000 vector instructions per stage and wave on top of the factor stage's ~650.
//
// Readers of (A, B) behind the start solve (only the constrained instances' QP kernels on the default path) get them from
// k_linearise_clist, which re-linearises those instances straight into their compact store (cfnmpc_kernels.hip).
#include <hip/hip_runtime.h>
#include <cstdio>

#include "cfnmpc_rg.hpp"

namespace cfn {

// what one fused stage reads from HBM (prefetched one stage ahead, while the previous stage factorises)
struct LfIn {
    double xr[10];  // x_k: q | v | w (external entries 3..12), replicated in every lane of the row
    double u[4];    // u_k, replicated
    double ua, yu;  // lanes a = L & 3: u_k[a], yref_k[13 + a]
    double xi, yk;  // lane i < 13: x_k[i], yref_k[i] (internal order)
    double xn;      // lane i < 13: x_{k+1}[i]
};
__device__ __forceinline__ void lf_load(const Params& P, const Lane& t, const int k, LfIn& in) {
    const int N = P.N;
    const gdouble* xb = blk(P.xit, t, N + 1, k, SZ_V13) + t.q * 13;
    SFOR(e, 3, 13, { in.xr[e - 3] = xb[int_of(e)]; });
    const gdouble* ub = blk(P.uit, t, P.N, k, SZ_V4) + t.q * 4;
    SFOR(a, 0, 4, { in.u[a] = ub[a]; });
    const int li = imin(t.L, 12), a = t.L & 3;
    in.ua = ub[a];
    in.xi = xb[li];
    in.xn = blk(P.xit, t, N + 1, k + 1, SZ_V13)[t.q * 13 + li];
    const gdouble* yb = blk(P.yref, t, N, k, SZ_Y) + t.q * 17;
    in.yk = yb[li];
    in.yu = yb[13 + a];
}

// Column role of a lane: state column with unit entry at external index `ecol` (lanes 0..9), or input column `cu`
// (lanes 10..13; 14 / 15 repeat input columns 0 / 1 and are never written to the tile).
struct LfLane {
    int ecol;        // external index of the unit entry of s(0); -1: input column
    double cj[4];    // df/du column of this lane divided by u_c: (2 KT, +-2 KA, +-2 KB, +-2 KC); 0 for state columns
    int cu;          // input column (0..3)
};
__device__ __forceinline__ LfLane lf_lane(const int L) {
    LfLane r;
    r.ecol = L < 3 ? 7 + L : (L < 7 ? L : (L < 10 ? L + 3 : -1));
    const int c = (L - 10) & 3;
    r.cu = c;
    const bool isu = L >= 10;
    const double sa = (c < 2) ? 1.0 : -1.0;             // w1 w2 | -w3 -w4      (cfnmpc_model.hpp: ju_col)
    const double sb = (c == 0 || c == 3) ? 1.0 : -1.0;  // w1 -w2 -w3 w4
    const double sc = (c == 0 || c == 2) ? 1.0 : -1.0;  // w1 -w2 w3 -w4
    r.cj[0] = isu ? 2.0 * KT : 0.0;
    r.cj[1] = isu ? 2.0 * KA * sa : 0.0;
    r.cj[2] = isu ? 2.0 * KB * sb : 0.0;
    r.cj[3] = isu ? 2.0 * KC * sc : 0.0;
    return r;
}

// Linearisation of one shooting interval in the row mapping: nominal RK4 (classic tableau, one step per interval)
// and this lane's sensitivity column, RK point by RK point.  col: the lane's column, EXTERNAL order.  The nominal
// slopes are the same in all 16 lanes of a row: lane 14 hands each point's slope to the row through `sb` (13 doubles)
// and every lane accumulates ITS OWN element only -- returns Phi[i] in lane i < 13 (internal order).
__device__ __forceinline__ double lf_linearise(const LfLane& r, const Lane& t, const double h, const LfIn& in,
                                               double (&col)[13], double* sb) {
    double rot[4], ju[4];
    {
        const double s1 = in.u[0] * in.u[0], s2 = in.u[1] * in.u[1], s3 = in.u[2] * in.u[2], s4 = in.u[3] * in.u[3];
        rot[0] = KT * (s1 + s2 + s3 + s4);
        rot[1] = KA * (s1 + s2 - s3 - s4);
        rot[2] = KB * (s1 - s2 - s3 + s4);
        rot[3] = KC * (s1 - s2 + s3 - s4);
        const double uc = r.cu == 0 ? in.u[0] : (r.cu == 1 ? in.u[1] : (r.cu == 2 ? in.u[2] : in.u[3]));
        SFOR(i, 0, 4, { ju[i] = r.cj[i] * uc; });
    }
    const int li = imin(t.L, 12);
    double xq[10], sq[10], acc[13];
    double ks = 0.0;   // sum of the nominal slopes, element li (internal order)
    SFOR(e, 0, 10, { xq[e] = in.xr[e]; sq[e] = (e + 3 == r.ecol) ? 1.0 : 0.0; });
    SFOR(e, 0, 13, { acc[e] = 0.0; });
    // (a run-time loop over the four RK points on purpose: unrolled, the scheduler interleaves the points -- the body-rate
    //  rows of all four depend on nothing else -- and the live ranges no longer fit two waves per SIMD)
#pragma unroll 1
    for (int p = 0; p < 4; p++) {
        const double wgt = (p == 0 || p == 3) ? 1.0 : 2.0;     // tableau weights 1 2 2 1
        const double cnh = (p == 2 ? 1.0 : 0.5) * h;           // node of the NEXT point (0.5, 0.5, 1) x h
        double dk[13], kk[13];
        lf_point(xq, sq, rot, ju, kk, dk);
        __syncthreads();
        if (t.L == 14) SFOR(rr, 0, 13, { sb[rr] = kk[ext_of(rr)]; });
        SFOR(e, 0, 13, { acc[e] += wgt * dk[e]; });
        SFOR(e, 0, 10, {
            sq[e] = ((e + 3 == r.ecol) ? 1.0 : 0.0) + cnh * dk[e + 3];
            xq[e] = in.xr[e] + cnh * kk[e + 3];
        });
        __syncthreads();
        ks += wgt * sb[li];
    }
    SFOR(e, 0, 13, { col[e] = ((e == r.ecol) ? 1.0 : 0.0) + (h / 6.0) * acc[e]; });
    return in.xi + (h / 6.0) * ks;
}

// Backward sweep over all N stages: linearise stage k in place, factorise it, prefetch stage k - 1 meanwhile.
__device__ __forceinline__ bool sweep_linfactor(const Params& P, const Lane& t, double* wt, double* sb, double* park,
                                                const double* qtab) {
    const int N = P.N;
    const double h = P.dt;
    double Pa[13];
    {   // terminal cost (as sweep_factor<true>)
        const double xN = ld13(blk(P.xit, t, N + 1, N, SZ_V13), t);
        const double yN = ld13(blk(P.yref_e, t, 1, 0, SZ_V13), t);
        double qv = 0.0;
        SFOR(j, 0, 13, { if (t.L == j) qv = P.WN[ext_of(j)] * (xN - yN); });
        SFOR(j, 0, 13, {
            const double qj = bc<j>(qv);
            Pa[j] = (t.L == j) ? P.WN[ext_of(j)] : ((t.L == 13) ? qj : 0.0);
        });
    }
    double wq = 0.0;
    SFOR(j, 0, 13, { if (t.L == j) wq = P.W[ext_of(j)]; });
    const double is13 = t.L == 13 ? 1.0 : 0.0;
    const int li = imin(t.L, 12);
    bool ok = true;
    LfIn in;
    lf_load(P, t, N - 1, in);
    for (int k = N - 1; k >= 0; k--) {
        StageIn<true> st;
        {
            // (the lane's column role is re-derived from an opaque copy of the lane index in every stage: hoisted out of
            //  the loop, its unit vector and df/du pattern would sit in ~40 registers for the whole sweep)
            int Lo = t.L;
            asm volatile("" : "+v"(Lo));
            const LfLane role = lf_lane(Lo);
            // the cost-to-go waits in LDS while the stage is linearised (13 registers the column needs)
            SFOR(j, 0, 13, { park[j * 64 + threadIdx.x] = Pa[j]; });
            double col[13];
            const double phil = lf_linearise(role, t, h, in, col, sb);
            // column form -> row form through the LDS tile (lane c writes column c as one contiguous run; lane i reads
            // element i of every column: consecutive lanes, consecutive banks -- the W transpose's own geometry)
            __syncthreads();
            if (t.L < 14) SFOR(rr, 0, 13, { wt[t.L * WT_ROW + rr] = col[ext_of(rr)]; });
            __syncthreads();
            SFOR(sl, 0, 10, { st.ar[sl] = wt[sl * WT_ROW + li]; });
            SFOR(a, 0, 4, { st.br[a] = wt[(10 + a) * WT_ROW + li]; });
            st.bv = phil - in.xn;
            st.qv = wq * (in.xi - in.yk);
            st.Rh = t.wu;
            st.g = t.wu * (in.ua - in.yu);
            SFOR(j, 0, 13, { Pa[j] = park[j * 64 + threadIdx.x]; });
        }
        // next stage's inputs: in flight while this stage factorises
        lf_load(P, t, imax(k - 1, 0), in);
        ok = factor_stage<true, false, false, true>(P, t, k, Pa, st, wq, is13, wt, sb, true, qtab) && ok;
        // checkpoints of the unconstrained cost-to-go (matrix part only), as k_factor leaves them (one store block with a
        // run-time checkpoint index: six specialised blocks keep six hoisted addresses alive through the whole sweep)
        int cidx = -1;
        SFOR(c, 0, N_CHK, { if (k == chk_stage(c)) cidx = c; });
        if (cidx >= 0) {
            gdouble* pc = gm(P.Pchk) + ((size_t)t.wave * N_CHK + cidx) * SZ_PP;
            SFOR(j, 0, 13, { if (t.L <= j) pc[pchk_col(j) + t.q * (j + 1) + t.L] = Pa[j]; });
        }
    }
    return ok;
}

KALIGN __global__ __launch_bounds__(64, 2) void k_linfactor(Params P) {
    __shared__ __attribute__((aligned(16))) double wtile[4][WT_TILE];
    __shared__ double btile[4][64];
    __shared__ double park[13 * 64];
    __shared__ __attribute__((aligned(16))) double qtab[16 * QT_ROW];
    const Lane t = lane_id(P);
    qtab_fill(P, qtab);
    __syncthreads();
    bool ok = sweep_linfactor(P, t, wtile[t.row], btile[t.row], park, qtab);
    ok = row_min(ok ? 1.0 : 0.0) > 0.0;
    if (t.L == 0 && t.valid) gm(P.status)[t.inst] = ok ? 0 : 4;
}

void launch_linfactor(const Params& P, hipStream_t st) {
    if (!P.v4b) { std::fprintf(stderr, "cfnmpc: k_linfactor needs the wave-blocked 4-vector layout (not a cond_N2 solver)\n"); return; }
    hipLaunchKernelGGL(k_linfactor, dim3(P.NW), dim3(64), 0, st, P);
}

}  // namespace cfn
