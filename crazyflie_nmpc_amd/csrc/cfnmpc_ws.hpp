// cfnmpc_ws.hpp -- device workspace description shared by the kernels and the C-ABI layer.
//
// v2 mapping ("row groups"): ONE NMPC INSTANCE PER 16-LANE DPP ROW, four instances per
// wavefront.  Lane i < 13 of a row owns ROW i of every 13-row object (P, A, B, K', W, ...) and
// element i of every 13-vector; lane 13 carries the affine part of the Riccati recursion as
// a 14th row (augmented recursion); lanes 14,15 idle.  Products are formed with
//     acc += own * row_newbcast<l>(src)          (v_mov_b64_dpp + v_fmac_f64)
// so no LDS / shuffles are needed for C = X * Y when X is row-distributed.
//
// Internal state order (lanes): p(0..2) v(3..5) q(6..9) w(10..12) -- i.e. the reference's
// order (acados_mpc.cpp:117-131) with the v and q blocks swapped, which makes
// A = dPhi/dx block UPPER triangular:
//            p  v  q  w
//        p [ I  *  *  * ]
//        v [ 0  *  *  * ]     97 stored entries
//        q [ 0  0  *  * ]
//        w [ 0  0  0  * ]
//
// HBM layout: wave-blocked.  For a field with S doubles per
// (wave, stage):   base + ((size_t)wave * stages + stage) * S + offset_in_block,
// and inside a block  [slot][instance-in-wave 0..3][contiguous lane range]  so that every
// load/store instruction of a wave touches one contiguous run of bytes with no padding.
#pragma once
#include <hip/hip_runtime.h>

namespace cfn {

// ---- internal <-> external state order -------------------------------------------------
__host__ __device__ constexpr int ext_of(int i) { return i < 3 ? i : (i < 6 ? i + 4 : (i < 10 ? i - 3 : i)); }
__host__ __device__ constexpr int int_of(int e) { return e < 3 ? e : (e < 7 ? e + 3 : (e < 10 ? e - 4 : e)); }

// ---- A in "row form" (AR): lane i holds A[i][3..12] in slots 0..9 (slot s <-> column s+3),
//      structurally zero slots are not stored.  Rows that own column c: c in v -> rows 0..5,
//      c in q -> rows 0..9, c in w -> rows 0..12  (a lane prefix).
__host__ __device__ constexpr int ar_n(int s) { return s < 3 ? 6 : (s < 7 ? 10 : 13); }
__host__ __device__ constexpr int ar_pre(int s) { return s < 3 ? 6 * s : (s < 7 ? 18 + 10 * (s - 3) : 58 + 13 * (s - 7)); }
static_assert(ar_pre(9) + ar_n(9) == 97, "97 stored entries of dPhi/dx");

// doubles per (wave, stage) block
constexpr int SZ_A = 4 * 97;   // AR: [slot][inst][lanes < ar_n(slot)]
constexpr int SZ_B = 4 * 52;   // BR: [a][inst][13]
constexpr int SZ_K = 4 * 52;   // KR (lane a holds K[a][0..12]): [l][inst][4]
constexpr int SZ_V13 = 4 * 13; // 13-vectors: [inst][13]
constexpr int SZ_V4 = 4 * 4;   // 4-vectors:  [inst][4]
constexpr int SZ_S = 4 * 10;   // packed symmetric 4x4: [inst][10]
constexpr int SZ_S4 = 4 * 16;  // full 4x4, row a in lane a: [c][inst][a]
constexpr int SZ_Y = 4 * 17;   // yref row: [inst][17] (first 13 in internal order)
constexpr int SZ_P = 4 * 13 * 13;  // a row-distributed 13x13: [col j][inst][13]
// checkpoints of the start solve (home blocks): the symmetric matrix packed -- column j keeps its rows 0..j only, [col j][inst][j + 1]
// (written for every instance and step, read by the constrained ones: 364 instead of 676 doubles per wavefront and checkpoint)
constexpr int SZ_PP = 4 * 91;
__host__ __device__ constexpr int pchk_col(int j) { return 2 * j * (j + 1); }                       // start of column j
__host__ __device__ inline int pchk_at(int j, int q, int r) {                                       // element (r, j), either triangle
    const int a = r < j ? r : j, b = r < j ? j : r;
    return 2 * b * (b + 1) + q * (b + 1) + a;
}
constexpr int SZ_PA = 4 * 13 * 14; // ... with the affine row (lane 13): [col j][inst][14]

constexpr int N_CHK = 6;  // Riccati checkpoints for the active-horizon QP (stage indices below)
// Compaction of the constrained rows: bins = head class (7: full horizon, 32, 24, 16, 12, 8, 4 -- longest first) x difficulty
// (N_DIFF classes of the number of violated inputs of the unconstrained minimiser, hardest first).  The four rows of a wave
// run their active-set solves in lock-step until the slowest has settled, so rows of similar difficulty should share a wave.
// Rounds 2 - 5 knew three classes (>= 4 | 2..3 | 1 violated inputs); round 6 splits the first (>= 20 | 9..19 | 4..8: mean
// solve counts 5.7 | 3.1 - 4.1 | 2.0 - 2.5 on the restatement's closed loop) for the MONOLITHIC active-set kernel (not the interior-point-only one: +1 %), i.e. where
// the constrained rows outnumber the SIMDs and the kernel is bound by the waves' work: under heavy disturbances most rows
// sit in that class (kicks x 2: 78 % of them; a wave spends 84 solve-stages where its rows need 58 -- 76 with the split) and
// k_as gets 8 % shorter (6.9 -> 6.3 ms; x 3: 17.9 -> 17.0 ms; bench workload: 0.636 -> 0.630 ms).  The solves + commit
// structure of the smaller fleets keeps the three classes: there every wave is resident at once, the kernel lasts as long as
// its hardest wave, and four hard rows in one wave restart their factorisations at the farthest stage ANY of them changed
// (+2 - 3 % at 32 768 / 40 960 instances with the split).  The violation count explains that much of the solve count and no
// more (correlation 0.74 - 0.78; eight classes: the same to 1 %); profiles/r06_notes.md section 13.
constexpr int N_DIFF = 5;
constexpr int N_BIN = (N_CHK + 1) * N_DIFF, BIN_STRIDE = 64;   // (bins per 64-instance group of the forward sweep, padded)
__host__ __device__ constexpr int diff_bin(int nviol, bool fine) {        // 0 = hardest
    return (fine && nviol >= 20) ? 0 : ((fine && nviol >= 9) ? 1 : (nviol >= 4 ? 2 : (nviol >= 2 ? 3 : 4)));
}
// Params.nipm: [0] length of the list, [1 + bin] rows per bin, and behind the bins:
constexpr int NI_LISTED = 60;   // rows k_ipm_list compacted for the interior-point fall-back (ilist2)
constexpr int NI_LONG16 = 61;   // listed rows with heads of more than 16 stages (first in the list)
constexpr int NI_LATE = 62;     // late rows of a split forward sweep (appended behind the list)
constexpr int NI_LONG24 = 63;   // listed rows with heads of more than 24 stages
static_assert(1 + N_BIN <= NI_LISTED && N_BIN <= BIN_STRIDE, "bin counts fit in front of the named slots of Params.nipm (64 ints)");
__host__ __device__ constexpr int chk_stage(int c) { return c == 0 ? 4 : (c == 1 ? 8 : (c == 2 ? 12 : (c == 3 ? 16 : (c == 4 ? 24 : 32)))); }

struct Params {
    int B, NW, N;  // instances, waves (= ceil(B/4)), horizon
    double dt;
    double W[17], WN[13];  // external order, as in cfnmpc_opts
    double u_min, u_max, tol, tau, thr0, lam0_min, mu0_scale;
    double clip_viol, clip_margin;   // interior point: clipped start (cfnmpc_opts.ipm_clip_viol / ipm_clip_margin)
    double as_skip_viol;             // rows beyond this many box widths skip the active-set iteration (cfnmpc_opts.as_skip_viol)
    int max_iter;
    int active_horizon;  // 1: interior-point sweeps only over the stages that can saturate
    double ah_margin;    // ... 'tight' = within this fraction of the box width of a bound
    int ah_extra;        // ... stages added behind the last tight stage
    // persistent iterate (acados nlp_out, acados_mpc.cpp:77) and per-step inputs
    double *xit;     // (N+1) stages x SZ_V13
    double *uit;     // N x SZ_V4
    double *xitn, *uitn;  // the NEW iterate of the running step (same shapes): every instance's step lands
                          // here (or its old iterate, if its QP failed); the host swaps old and new afterwards
    double *x0;      // 1 x SZ_V13
    double *yref;    // N x SZ_Y
    double *yref_e;  // 1 x SZ_V13
    // linearisation
    int ab16;    // AR, BR, b of THIS parameter set are grouped by 16 blocks: [group][stage][block of the group][SZ] (the home fields; 0: [block][stage][SZ], the compact store)
    double *AR;  // N x SZ_A   A row-distributed (lane i holds A[i][3..12])
    double *BR;  // N x SZ_B   B row-distributed (lane i holds B[i][0..3])
    double *b;        // N x SZ_V13
    // Riccati factors
    double *KR;       // N x SZ_K   gain, lane a < 4 holds K[a][0..12]
    double *Sinv;     // N x SZ_S
    double *d;        // N x SZ_V4
    double *Pchk;     // N_CHK x SZ_PP (home: packed) / N_CHK x SZ_P (compact copy cPchk)   cost-to-go of the unconstrained tail at the checkpoints
    // interior-point state, N x SZ_V4 each
    double *v, *tl, *tu, *ll, *lu, *rg, *dva, *dvc, *Rh, *g;
    // compact scratch of the interior-point kernel (same block shapes, indexed by compacted slot)
    double *cAR, *cBR, *cKR, *cSinv, *cd, *cPchk, *cv, *cuit;
    double *cbv;     // compact b (N x SZ_V13 per compact block): only the fused start solve's k_linearise_clist fills it
    // active-set solve, per compact block and stage: G = B'PA in the gain layout (SZ_K), the rows of
    // S = R^ + B'PB (SZ_S4: [c][inst][a]) and rho = B'(p + P b_eff) (SZ_V4)
    double *cGR, *cS, *crho;
    double *cPs;     // AS_PSAVE x SZ_PA per compact block: cost-to-go (with affine row) of the stages 1..31
    int active_set;  // 1: try the primal-dual active-set solve before the interior-point iteration
    // warm start of the active set (cfnmpc_opts.as_warm): per instance the classification its last SETTLED active-set solve ended
    // with (wcls [inst][N * 4] bytes: 0 free, 1 lower, 2 upper; 0 behind that solve's head) and whether the instance's previous RTI
    // step ended that way (wvalid [inst]: the forward sweep clears it for feasible instances, the QP kernels set / clear it)
    int as_dense;                // 1: rows with heads of at most 16 stages are solved by k_as_dense (cfnmpc_asdense.hip), beside k_as_solves
    // host-only (never read on the device): side stream + fork / join events on which k_as_solves -- the few rows with longer
    // heads, a latency chain of its own -- runs BESIDE k_as_dense (hipStream_t / hipEvent_t, owned by the solver)
    void *as_side, *as_fork, *as_join;
    void *as_side2, *as_join2;   // ... and the second part of the split forward sweep beside both
    int as_range;                // k_as_solves beside the dense kernel: 0 = every row with a head of more than 16 stages; 1 (behind a split
                                 // sweep) = heads of 24 stages only -- they read nothing behind stage 24 and run beside part two of the
                                 // sweep; the rows with longer heads (classes 32 / N) are left to the retry kernel (k_scatter flags them)
    // split forward sweep (cfnmpc_opts.forward_split): 0 = one launch; H = 24: k_forward_p1 over [0, H) -> compaction -> k_forward_p2 over
    // [H, N) beside the constrained rows' kernels; hand-over [group of 64][13 | 4][64]
    int fwd_split;
    double* fs_dx;
    int* fs_st;
    int as_warm;
    unsigned char* wcls;
    int* wvalid;
    int *status, *iters, *head;  // per instance (head: stages the interior-point sweeps cover, 0 = none)
    double *res, *viol;          // per instance
    int ipm_listed;              // 1: k_ipm_rest works on ilist2 (fleets whose fall-back rows may exceed one wave per SIMD); 0: on ilist, filtered
    int *ilist2;                 // the rows of ilist the active-set kernels left for the interior point (k_ipm_list; count in nipm[NI_LISTED])
    int *ilist;                  // compacted list of the instances that need the interior-point method
    int *nipm;                   // [0] its length, [1 + bin] instances per compaction bin
    int *blkcnt;                 // per 64-instance group of k_forward: instances per compaction bin [group][BIN_STRIDE]
    int *done;                   // per instance: 1 = finished by the active-set kernel (k_as), 0 = left for k_ipm_rest
    int *rank;                   // per instance: (bin << 8) | rank among the same-bin instances of its group
    // partial condensing (cfnmpc_opts.cond_N2 < N): the N stages are regrouped into cond_N2 blocks,
    // the first cond_rem of them cond_M + 1 stages long, the others cond_M (cond_N2 = 0: off)
    // level-synchronous active-set passes (cfnmpc_opts.as_pipeline; k_asp / k_ascommit in cfnmpc_kernels.hip)
    int as_passes;               // 0: monolithic k_as; p > 0: p single-solve launches + one launch for the remaining solves
    int as_grid;                 // wavefronts per pass launch (grid-stride over the work list)
    int as_sparse_max;           // active-set kernels: one list slot per wave while the list is no longer than this (0: never)
    int *aslist, *ascnt;         // work lists [set 0..2][bin 0..6][compact slots], their lengths [set][bin]
    int *askst, *asst, *asok;    // per compact slot: restart stage of its next solve; 1 = settled (to be committed); last factorisation positive definite
    double *czdx;                // per compact slot: dx_k of its last solve, (N + 1) x 13
    // per-stage, per-input input box (cfnmpc_set_box_stages; NULL: the scalar box u_min / u_max): instance-major
    // [inst][stage][4] like uit, and the compact copies of the constrained-QP kernels
    double *lbs, *ubs, *clbs, *cubs;
#ifdef CFN_DEV
    // stage-chunked hand-over experiment (cfnmpc_debug_chunked_pair): stage ranges of k_linearise / k_factor_chunk
    int lin_k0, lin_k1, fk_lo, fk_hi;
    double *Ppark;               // cost-to-go between the chunks, [wave][13][64]
#endif
#ifdef CFN_DEV
    int forward_half;            // 1: k_forward_half (two waves per SIMD) instead of k_forward (sub-fleet experiment)
#endif
    int v4b;                     // layout of the HOME 4-vectors (uit, uitn, d, v, lbs, ubs): 1 = wave-blocked [wave][stage][inst & 3][4] (the four
                                 // instances of a block share one 128-byte line per stage, consumed whole by a wavefront in either mapping),
                                 // 0 = instance-major [inst][stage][4] (the compact copies always; the home arrays of the partial-condensing path)
    int forward_rg;              // 1: forward sweep of the start solve on the stored blocks (k_forward_rg; small fleets)
    int clist_chunks;            // workgroups per 64-slot group of k_linearise_clist (stage chunks)
    int fused;                   // start solve: 0 = k_linearise + k_factor on stored (A, B, b); 1 = k_linfactor, nothing stored (the QP
                                 // kernels of the constrained instances read the compact store k_linearise_clist fills);
                                 // 2 = k_linfactor for the factorisation, stored blocks for everything else (validation)
    int cond_N2, cond_M, cond_rem;
    double *cb;                  // condensed blocks, [instance][block][cb_size(w_max)] (layout: cfnmpc_pcond.hip)
};

// The linearisation (AR, BR, b) of the HOME blocks: [group of 16 blocks][stage][block of the group][sz] -- a lane-per-instance wave
// (64 instances = one group) then writes one 50 / 27 / 7 KB run per stage and field instead of 16 runs 155 KB apart (49 000 concurrent
// write streams at 65 536 instances: k_linearise then took 1.0 or 1.25 ms from one process to the next on the same box,
// profiles/r05_linearise_stores.md), and the 16 row-group waves of a group walk the same 50 KB windows.  The compact store keeps
// [block][stage][sz] (Params.ab16 = 0 in the kernels' compact parameter sets).
__device__ __forceinline__ size_t abidx(const Params& P, int wave, int k) {
    return P.ab16 ? ((size_t)(wave >> 4) * P.N + k) * 16 + (wave & 15) : (size_t)wave * P.N + k;
}

// ---- partial condensing geometry --------------------------------------------------------------
constexpr int COND_MMAX = 10;   // longest block supported (4 * 10 = 40 condensed inputs)
__host__ __device__ inline int cond_len(const Params& P, int j) { return j < P.cond_rem ? P.cond_M + 1 : P.cond_M; }
__host__ __device__ inline int cond_start(const Params& P, int j) {
    return j < P.cond_rem ? j * (P.cond_M + 1) : P.cond_rem * (P.cond_M + 1) + (j - P.cond_rem) * P.cond_M;
}
__host__ __device__ inline int cond_mmax(const Params& P) { return P.cond_M + (P.cond_rem ? 1 : 0); }
// condensed block of m stages: z = (U (4m), dx (13), 1), w = 4m + 14 entries;
//   H  : symmetric w x w, packed lower (row r, col c <= r at r (r + 1) / 2 + c)   -- cost 1/2 z'H z
//   D  : 13 x w row-major                                                        -- dx+ = D z
__host__ __device__ constexpr int cond_w(int m) { return 4 * m + 14; }
__host__ __device__ constexpr int cond_tri(int w) { return w * (w + 1) / 2; }
__host__ __device__ constexpr int cb_size(int mmax) { return cond_tri(cond_w(mmax)) + 13 * cond_w(mmax); }

// linearisation of all instances / of the instances in P.ilist; `chunks` = number of workgroups the
// (independent) shooting intervals of one 64-instance group are spread over
void launch_linearise(const Params& P, int chunks, hipStream_t st);
void launch_linearise_list(const Params& P, int chunks, hipStream_t st);
// fused start solve: the instances of P.ilist (which = 0) / P.ilist2 (1) re-linearised into the compact store
void launch_linearise_clist(const Params& P, int chunks, int which, hipStream_t st);
// ev (optional): four events recorded after k_factor / the forward sweep / the compaction / the active-set kernels
void launch_qp(const Params& P, hipStream_t st, hipEvent_t* ev = nullptr);        // = launch_qp_start + launch_qp_ipm
void launch_qp_start(const Params& P, hipStream_t st, hipEvent_t* ev = nullptr, bool skip_factor = false);  // factor, forward (+ full step of feasible rows), compact
void launch_qp_ipm(const Params& P, hipStream_t st, hipEvent_t* ev = nullptr);    // constrained instances (+ their full step)
// partial condensing path (cfnmpc_pcond.hip): pcond -> condensed Riccati -> forward sweep with expand ->
// interior point on the condensed QP for the constrained instances
void launch_pcond(const Params& P, hipStream_t st);
void launch_cfactor(const Params& P, hipStream_t st);
#ifdef CFN_DEV
void launch_factor_chunk(const Params& P, hipStream_t st);   // experiment: stages [P.fk_lo, P.fk_hi) of the start solve
#endif
void launch_factor_only(const Params& P, hipStream_t st);
void launch_as_dense(const Params& P, int grid, hipStream_t st);   // cfnmpc_asdense.hip: head-condensed dense active-set solves
void launch_linfactor(const Params& P, hipStream_t st);    // cfnmpc_linfactor.hip: fused linearisation + backward factorisation
void launch_cforward(const Params& P, hipStream_t st);   // (in cfnmpc_kernels.hip: k_forward with condensed gains)
void launch_cipm(const Params& P, hipStream_t st);
void launch_qp_cond(const Params& P, hipStream_t st);    // = the four above
void launch_estimate(int B, const double* meas, double* filt, const double* u, double dt, int use_lpf, double delay,
                     int steps, double* x_est, double* x_pred, hipStream_t st);
void launch_sim(int B, const double* x, const double* u, double T, int steps, double* xn, hipStream_t st);
// AoS [B][S][E] (external order) <-> wave-blocked vectors; perm13: first 13 entries of each
// row are states and are permuted to the internal order.
void launch_put(int B, int S, int E, int perm13, const double* aos, double* blk, hipStream_t st, int v4b = 0);   // v4b: E = 4 fields in the wave-blocked layout (Params.v4b)
void launch_get(int B, int S, int E, int perm13, int s0, int Stot, const double* blk, double* aos, hipStream_t st, int v4b = 0);
void launch_windows(const Params& P, const double* traj, int n_rows, int* mode, int* iter, const double* des,
                    double uss, hipStream_t st);
void launch_init_iterate(const Params& P, int mode, hipStream_t st);
void launch_reinit_failed(const Params& P, hipStream_t st);   // cfnmpc_opts.reinit_failed
// output stage of the reference node for the fleet (cmd_vel [B][4] doubles, motvel [B][4] int32 or NULL)
void launch_postproc(const Params& P, double* cmd_vel, int* motvel, hipStream_t st);

}  // namespace cfn
