// cfnmpc_api.cpp -- C-ABI of the batch engine (include/cfnmpc.h) over the HIP kernels.
// Host side only: allocation of the SoA workspace, AoS<->SoA staging, kernel launches.
// There is no CPU fallback: without a usable HIP device cfnmpc_create() fails.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <utility>
#include <vector>

#include "../../include/cfnmpc.h"
#include "cfnmpc_model.hpp"
#include "cfnmpc_ws.hpp"
#ifdef CFN_DEV
#include "cfnmpc_dev.h"
#endif

struct cfnmpc_solver {
    cfn::Params P;
    int device;
    std::vector<void*> allocs;
    double* stage_buf;  // device staging buffer for AoS transfers (largest AoS array)
    size_t stage_doubles;
    // cfnmpc_step_host: pinned host mirror + device buffers of one step's inputs / outputs (lazy)
    double *h_io, *d_io;
    size_t io_doubles;
    unsigned long long bytes;
    // optional per-kernel timing with HIP events on the launch stream (cfnmpc_set_profiling)
    int profiling;
    std::vector<hipEvent_t> ev;  // EV_PER_STEP events per timed RTI step: start | linearise | factor | forward | compaction | active set | end
    size_t ev_used;
    // preparation-phase overlap (development builds, CFNMPC_OVERLAP=1; always 0 in the product): the linearisation for the NEXT
    // step is written to the alternate (AR, BR, b) set while the interior-point kernel still
    // reads the current one
    int overlap;
    double *AR2, *BR2, *b2;
    hipStream_t aux;             // low-priority stream of the early linearisation pass
    hipEvent_t ev_start, ev_aux; // start solve done (on the caller's stream) / early pass done (on aux)
    bool lin_valid;              // (P.AR, P.BR, P.b) hold the linearisation of the current iterate
    int chunks_all, chunks_list; // workgroups per 64-instance group in the two linearisation passes
    // captured step (cfnmpc_opts.step_graph): the launches of one RTI step as a hipGraph, one per parity of
    // the iterate buffers (kernel arguments are by value and the host swaps xit / xitn after every step)
    int use_graph;
    hipStream_t cap;             // capture stream
    hipGraphExec_t gexec[2];
    hipEvent_t glaunched[2];     // recorded behind the last launch of gexec[p]: waited for before that exec is destroyed
    bool gvalid[2];
    int parity;                  // which of the two argument sets the NEXT step uses
    int reinit_failed;           // cfnmpc_opts.reinit_failed
    double *lbs_keep, *ubs_keep; // per-stage boxes (cfnmpc_set_box_stages), allocated at the first call
    double* box_blk[4];          // ... the four blocks behind them (home lb / ub, compact lb / ub); a failed attempt keeps what it got
};

namespace {

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            std::fprintf(stderr, "cfnmpc: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_),   \
                         __FILE__, __LINE__);                                                       \
            return CFNMPC_EHIP;                                                                     \
        }                                                                                           \
    } while (0)

// The solver lives on the device that was current at cfnmpc_create; every entry point that touches
// it makes that device current for the duration of the call (a caller that drives several GPUs
// from one thread may have another one selected).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(const cfnmpc_solver* s) {
        if (s && hipGetDevice(&prev) == hipSuccess && prev != s->device) switched = hipSetDevice(s->device) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

constexpr size_t MAX_PROFILED_STEPS = 4096;   // cfnmpc_get_profile resets the count
constexpr size_t EV_PER_STEP = 7;
// Kernel choices that depend on how the fleet fills the device, in units of its SIMDs (S = 4 x compute units = wave slots at
// one wave per SIMD; MI355X: 1024) -- measured per horizon, profiles/r04_thresholds.md (tools/threshold_sweep.py):
//   * forward sweep on the stored blocks (row groups, B / 4 waves) instead of the matrix-free one (B / 64 waves of N
//     sequential stages): while B / 4 waves fit about twice on the SIMDs; the row-group sweep streams 240 N doubles per
//     instance, so long horizons leave it earlier (cross-over 7000 instances at N = 30 and 50, 5500 at N = 100);
//   * active-set solves + commit kernel instead of the monolithic kernel: while the ~8 % constrained rows make fewer
//     waves than there are SIMDs by a margin (roll-out = latency chain); re-measured after the 4-vector layout change:
//     N = 50: -3.5 % at 24 S, -0.5 % at 32 S, equal at 48 S; N = 100: -3.6 % up to 32 S, equal at 48 S; N = 30: +4.7 % from 24 S on;
//   * fall-back rows compacted before the interior point: from 16 S instances;
//   * head-condensed dense active-set solves (k_as_dense) for the rows with heads of at most 16 stages: below 18 S instances
//     (round 5, one box, A/B per size: 2 S +20.6 %, 4 S +8.5 %, 8 S +9.6 %, 12 S +1.8 %, 16 S +1.1 %, 20 S -1.8 %, 24 S -1.4 %,
//     28 S -0.5 %, 32 S -5 %: one row per wavefront and SIMD -- while the ~8 % constrained rows fit the SIMDs once the kernel lasts
//     as long as its hardest row, 50 - 90 us instead of 220; beyond that its ~47 us per row are rounds of waves and the four
//     rows per wavefront of k_as_solves / k_as win on throughput), profiles/r05_as_dense.md;
//   * split forward sweep (k_forward_p1 | k_forward_p2 beside the constrained rows' kernels): wherever that structure runs with the
//     matrix-free sweep (one box: 6144 instances +3.5 % against the unsplit matrix-free sweep and +2 % against the row-group one,
//     8192: +7.2 %, 12 288: +1.2 %, 16 384: +0.7 %; at 4096 the row-group sweep stays ahead, 7.33 against 6.75 M).
struct Choice { bool forward_rg, as_commit, ipm_listed, as_dense, forward_split; };
inline Choice choose_kernels(int batch, int N, int simds) {
    const long S = simds > 0 ? simds : 1024;
    Choice c;
    c.forward_rg = (long)batch < 6 * S;   // (round 5: 8 S -> 6 S for N <= 64 too -- the split matrix-free sweep beats the row-group one from 6 S on: 9.37 against 9.18 M at 6144 instances)
    c.as_commit = (long)batch < (N <= 40 ? 20 : 36) * S;
    c.ipm_listed = (long)batch >= 16 * S;
    c.as_dense = (long)batch < 18 * S;
    c.forward_split = true;
    return c;
}

// `on_device` argument: 0 host (synchronous), 2 host (enqueued only), anything else: device pointer
inline bool is_host(int on_device) { return on_device == CFNMPC_ON_HOST || on_device == CFNMPC_ON_HOST_ASYNC; }

template <typename T>
int dev_alloc(cfnmpc_solver* s, T** p, size_t count) {
    void* q = nullptr;
    if (hipMalloc(&q, count * sizeof(T)) != hipSuccess) return CFNMPC_ENOMEM;
    s->allocs.push_back(q);   // (before the memset: cfnmpc_free releases it on any later failure)
    if (hipMemset(q, 0, count * sizeof(T)) != hipSuccess) return CFNMPC_EHIP;
    s->bytes += count * sizeof(T);
    *p = static_cast<T*>(q);
    return CFNMPC_OK;
}

// copy an AoS array [B][S][E] (external state order) from the caller into a workspace field
int put_field(cfnmpc_solver* s, const double* src, int on_device, int S, int E, int perm13, double* field,
              hipStream_t st) {
    const cfn::Params& P = s->P;
    const size_t n = (size_t)P.B * S * E;
    const double* dsrc = src;
    const bool host = is_host(on_device);
    if (host) {
        if (n > s->stage_doubles) return CFNMPC_EINVAL;
        HIP_TRY(hipMemcpyAsync(s->stage_buf, src, n * sizeof(double), hipMemcpyHostToDevice, st));
        dsrc = s->stage_buf;
    }
    cfn::launch_put(P.B, S, E, perm13, dsrc, field, st, E == 4 ? P.v4b : 0);
    HIP_TRY(hipGetLastError());
    // the staging buffer is reused: wait, unless the caller orders the next use on the same stream itself
    if (on_device == CFNMPC_ON_HOST) HIP_TRY(hipStreamSynchronize(st));
    return CFNMPC_OK;
}

// copy stages s0..s0+S-1 of a workspace field into the caller's AoS array [B][S][E]
int get_field(cfnmpc_solver* s, double* dst, int on_device, int S, int E, int perm13, int s0, int Stot,
              const double* field, hipStream_t st) {
    const cfn::Params& P = s->P;
    const size_t n = (size_t)P.B * S * E;
    double* ddst = dst;
    const bool host = is_host(on_device);
    if (host) {
        if (n > s->stage_doubles) return CFNMPC_EINVAL;
        ddst = s->stage_buf;
    }
    cfn::launch_get(P.B, S, E, perm13, s0, Stot, field, ddst, st, E == 4 ? P.v4b : 0);
    HIP_TRY(hipGetLastError());
    if (host) {
        HIP_TRY(hipMemcpyAsync(dst, ddst, n * sizeof(double), hipMemcpyDeviceToHost, st));
        if (on_device == CFNMPC_ON_HOST) HIP_TRY(hipStreamSynchronize(st));
    }
    return CFNMPC_OK;
}

// Q may be positive SEMI-definite (the reference's dynamic_reconfigure ranges start at 0.0,
// config/crazyflie_params.cfg:19-35): state weights >= 0, input weights > 0 (R must be definite:
// S = R + B'PB is what the Riccati recursion inverts); no NaN.
bool weights_ok(const double* W, const double* WN) {
    if (W) {
        for (int i = 0; i < 13; i++) if (!(W[i] >= 0.0)) return false;
        for (int i = 13; i < 17; i++) if (!(W[i] > 0.0)) return false;
    }
    if (WN) for (int i = 0; i < 13; i++) if (!(WN[i] >= 0.0)) return false;
    return true;
}

}  // namespace

extern "C" {

const char* cfnmpc_version(void) { return "cfnmpc 0.9 (gfx950; row-group Riccati with fused DPP broadcast FMAs, lane-per-instance linearisation and matrix-free forward sweep, primal-dual active-set QP solves; options: fused start solve (linearisation inside the factorisation), partial condensing, small-fleet forward sweep; device output stage; multi-GPU shards)"; }

void cfnmpc_default_opts(cfnmpc_opts* o) {
    // generate_c_code.py:41-42,63-84,109,133-134
    static const double W[CFNMPC_NY] = {120.0, 100.0, 100.0, 1e-3, 1e-3, 1e-3, 1e-3, 0.7, 1.0,
                                        4.0,   1e-5,  1e-5,  10.0, 0.06, 0.06, 0.06, 0.06};
    o->struct_size = (int)sizeof(cfnmpc_opts);
    o->N = 50;
    o->dt = 0.75 / 50;
    for (int i = 0; i < CFNMPC_NY; i++) o->W[i] = W[i];
    for (int i = 0; i < CFNMPC_NYN; i++) o->WN[i] = 50.0 * W[i];
    o->u_min = 0.0;
    o->u_max = 22.0;
    o->tol = 1e-8;
    o->max_iter = 50;
    o->tau = 0.995;
    o->thr0 = 1.0;
    o->lam0_min = 1e-2;
    o->mu0_scale = 0.1;
    o->active_horizon = 1;
    o->ah_margin = 0.10;
    o->ah_extra = 4;
    o->active_set = 1;
    o->forward_sweep = 0;
    o->cond_N2 = 0;
    o->step_graph = 0;
    o->as_passes = 0;
    o->ipm_clip_viol = 2.0;
    o->ipm_clip_margin = 0.05;
    o->as_skip_viol = 4.0;
    o->reinit_failed = 0;
    o->start_solve = 0;
    o->as_warm = 0;
    o->as_dense = 0;
    o->forward_split = 0;
}

int cfnmpc_default_opts_v(cfnmpc_opts* o, int sizeof_opts) {
    if (!o || sizeof_opts != (int)sizeof(cfnmpc_opts)) return CFNMPC_EINVAL;   // nothing is written into a struct of another size
    cfnmpc_default_opts(o);
    return CFNMPC_OK;
}
int cfnmpc_opts_size(void) { return (int)sizeof(cfnmpc_opts); }
int cfnmpc_abi_version(void) { return CFNMPC_ABI_VERSION; }

int cfnmpc_create(cfnmpc_solver** out, int batch, const cfnmpc_opts* opts) {
    if (!out || batch <= 0) return CFNMPC_EINVAL;
    // ABI guard: the first field is the size the caller's filler saw -- read BEFORE the struct is copied
    if (opts && opts->struct_size != (int)sizeof(cfnmpc_opts)) {
        std::fprintf(stderr, "cfnmpc: cfnmpc_opts of %d bytes handed to a library built for %d (ABI %d): rebuild against include/cfnmpc.h\n",
                     opts->struct_size, (int)sizeof(cfnmpc_opts), CFNMPC_ABI_VERSION);
        return CFNMPC_EINVAL;
    }
    cfnmpc_opts o;
    if (opts) o = *opts; else cfnmpc_default_opts(&o);
    // Overlapped preparation (the next step's linearisation beside the constrained rows' kernels, double-buffered A / B / b):
    // measured slower at every fleet size -- since round 6 an experiment of the development build
    // (make DEV=1; CFNMPC_OVERLAP=1 in the environment), not an option of the product.
    int overlap_linearise = 0;
#ifdef CFN_DEV
    if (const char* e = std::getenv("CFNMPC_OVERLAP")) overlap_linearise = std::atoi(e) ? 1 : 0;
#endif
    if (o.N < 5 || o.N > 4096 || !(o.dt > 0) || !(o.u_max > o.u_min) || o.max_iter < 0 ||
        !(o.ah_margin >= 0.0 && o.ah_margin < 0.5) || o.ah_extra < 0) return CFNMPC_EINVAL;
    // QP parameters that would otherwise only show up as NaN / status 4 at run time
    if (!(o.tol > 0.0) || !(o.tau > 0.0 && o.tau < 1.0) || !(o.thr0 > 0.0) || !(o.lam0_min > 0.0) ||
        !(o.mu0_scale >= 0.0) || !(o.ipm_clip_viol >= 0.0) || !(o.ipm_clip_margin > 0.0 && o.ipm_clip_margin < 0.5) ||
        !(o.as_skip_viol >= 0.0)) return CFNMPC_EINVAL;
    if (!weights_ok(o.W, o.WN)) return CFNMPC_EINVAL;
    // partial condensing: cond_N2 blocks of at most COND_MMAX stages; not combined with the overlapped preparation
    if (o.cond_N2 < 0 || o.cond_N2 > o.N) return CFNMPC_EINVAL;
    const int cond_N2 = (o.cond_N2 == 0 || o.cond_N2 == o.N) ? 0 : o.cond_N2;
    if (cond_N2 && ((o.N + cond_N2 - 1) / cond_N2 > cfn::COND_MMAX || overlap_linearise)) return CFNMPC_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        std::fprintf(stderr, "cfnmpc: no HIP device available (this library has no CPU path)\n");
        return CFNMPC_EHIP;
    }
    cfnmpc_solver* s = new (std::nothrow) cfnmpc_solver();
    if (!s) return CFNMPC_ENOMEM;
    s->bytes = 0;
    s->h_io = s->d_io = nullptr;
    s->io_doubles = 0;
    s->profiling = 0;
    s->ev_used = 0;
    s->overlap = overlap_linearise ? 1 : 0;
    s->AR2 = s->BR2 = s->b2 = nullptr;
    s->aux = nullptr;
    s->ev_start = s->ev_aux = nullptr;
    s->lin_valid = false;
    s->use_graph = o.step_graph ? 1 : 0;
    s->cap = nullptr;
    s->gexec[0] = s->gexec[1] = nullptr;
    s->glaunched[0] = s->glaunched[1] = nullptr;
    s->gvalid[0] = s->gvalid[1] = false;
    s->parity = 0;
    s->lbs_keep = s->ubs_keep = nullptr;
    for (double*& q : s->box_blk) q = nullptr;
    s->reinit_failed = o.reinit_failed ? 1 : 0;
    int simds = 1024;   // SIMDs of the device the solver is created on
    {
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) simds = 4 * prop.multiProcessorCount;
    }
    const Choice pick = choose_kernels(batch, o.N, simds);
    // the shooting intervals of a 64-instance group are independent: spread them over enough
    // workgroups to fill the 1024 SIMDs when the batch alone does not (a single instance then
    // linearises its 50 intervals in parallel instead of one after the other)
    {
        // k_linearise runs one wavefront per SIMD: a launch of groups x c workgroups takes ceil(groups c / SIMDs) rounds of
        // ceil(N / c) stages each (+ a prologue per workgroup) -- take the c that minimises it (rounding groups c UP past the
        // number of SIMDs would cost a whole second round: 255 groups x 5 chunks = 1275 workgroups ran 35 % slower than x 4)
        const int groups = (batch + 63) / 64;
        int c = 1;
        if (s->overlap) {
            c = 5;
            if (groups * c < simds) c = (simds + groups - 1) / groups;
        } else {
            double best = 1e300;
            for (int cc = 1; cc <= o.N; cc++) {
                const long rounds = ((long)groups * cc + simds - 1) / simds;
                const double cost = (double)rounds * ((o.N + cc - 1) / cc + 0.35);   // 0.35 stage-equivalents of prologue per workgroup
                if (cost < best - 1e-9) { best = cost; c = cc; }
            }
        }
        s->chunks_all = c < 1 ? 1 : (c > o.N ? o.N : c);
    }
    s->chunks_list = 10;
#ifdef CFN_DEV   // development builds only (make DEV=1): the shipped library reads no environment variables
    if (const char* e = std::getenv("CFNMPC_LIN_CHUNKS")) {
        int a = 0, b = 0;
        if (std::sscanf(e, "%d,%d", &a, &b) == 2 && a > 0 && b > 0) { s->chunks_all = a; s->chunks_list = b; }
    }
#endif
    if (hipGetDevice(&s->device) != hipSuccess) { delete s; return CFNMPC_EHIP; }
    cfn::Params& P = s->P;
    std::memset(&P, 0, sizeof P);
    P.B = batch;
    P.NW = (batch + 3) / 4;
    P.N = o.N;
    P.dt = o.dt;
    for (int i = 0; i < 17; i++) P.W[i] = o.W[i];
    for (int i = 0; i < 13; i++) P.WN[i] = o.WN[i];
    P.u_min = o.u_min; P.u_max = o.u_max; P.tol = o.tol; P.tau = o.tau; P.thr0 = o.thr0;
    P.lam0_min = o.lam0_min; P.mu0_scale = o.mu0_scale; P.max_iter = o.max_iter;
    P.clip_viol = o.ipm_clip_viol; P.clip_margin = o.ipm_clip_margin; P.as_skip_viol = o.as_skip_viol;
    P.active_horizon = o.active_horizon ? 1 : 0;
    P.ah_margin = o.ah_margin;
    P.ah_extra = o.ah_extra;
    P.active_set = o.active_set ? 1 : 0;
    P.as_warm = (o.as_warm && o.active_set) ? 1 : 0;
    if (o.forward_sweep < 0 || o.forward_sweep > 2 || (o.step_graph && overlap_linearise) ||
        (o.reinit_failed && overlap_linearise)) { delete s; return CFNMPC_EINVAL; }
#ifdef CFN_DEV   // (development builds: also -2 and 1..12, the instance-contiguous store / level-synchronous passes of round 3)
    if (o.as_passes < -3 || o.as_passes > 12) { delete s; return CFNMPC_EINVAL; }
#else
    if (o.as_passes != 0 && o.as_passes != -1 && o.as_passes != -3) { delete s; return CFNMPC_EINVAL; }
#endif
    // internal: 0 = monolithic k_as, -1 = every solve in one launch on the compact z store + commit, -2 = the monolithic
    // kernel's solves + commit, p > 0 = p single-solve passes
    P.as_passes = o.as_passes > 0 ? o.as_passes : (o.as_passes == -2 ? -1 : (o.as_passes == -3 ? -2 : 0));
    if (o.as_passes == 0 && pick.as_commit && o.start_solve != 2) P.as_passes = -2;   // small fleets: solves + commit kernel (measured); the commit kernel reads stored blocks
#ifdef CFN_DEV
    if (const char* e = std::getenv("CFNMPC_AS_PASSES")) {   // (internal encoding)
        const int v = std::atoi(e);
        if (v >= -2 && v <= 12) P.as_passes = v;
    }
#endif
    {   // pass launches: two wavefronts per SIMD of this device
        P.as_grid = 2 * simds;
#ifdef CFN_DEV
        if (const char* e = std::getenv("CFNMPC_AS_GRID")) { const int v = std::atoi(e); if (v > 0) P.as_grid = v; }
#endif
    }
    // one fall-back row per wave is as fast as four while those waves fit one per SIMD (1024 rows: 1.6 % of 65 536 instances,
    // 2.5 % fall back at three times the bench's disturbances): only large fleets pay the compaction's extra launch
    P.as_sparse_max = P.as_grid / 2;   // = the SIMDs of the device: one constrained row per wave while they all fit at once
    P.ipm_listed = pick.ipm_listed ? 1 : 0;
    // head-condensed dense active-set solves (cfnmpc_asdense.hip) beside the solves + commit structure; scalar box, stored blocks
    // (every option is validated BEFORE the side streams below are created: the refusals here own nothing but `s`)
    if (o.as_dense < -1 || o.as_dense > 1 || o.start_solve < 0 || o.start_solve > 3 || o.forward_split < -1 || o.forward_split > 1) { delete s; return CFNMPC_EINVAL; }
    if (o.as_dense == 1 && (P.as_passes != -2 && o.as_passes != 0)) { delete s; return CFNMPC_EINVAL; }
    if (o.as_dense == 1 && o.as_passes == 0 && o.start_solve != 2 && !cond_N2) P.as_passes = -2;   // asked for: the structure it lives in
    P.as_dense = (P.as_passes == -2 && P.active_set && !P.as_warm && o.as_dense != -1 && (o.as_dense == 1 || pick.as_dense)) ? 1 : 0;
    // an EXPLICIT request that cannot be honoured is refused, not dropped (as_dense = 1 beside start_solve = 2, cond_N2, as_warm or
    // active_set = 0; forward_split = 1 outside the dense structure, with the row-group sweep, short horizons or the overlapped
    // preparation -- whose early pass would read the iterate while part two of the sweep is still writing it)
    if (o.as_dense == 1 && !P.as_dense) { delete s; return CFNMPC_EINVAL; }
    const bool split_ok = P.as_dense && !cond_N2 && o.start_solve != 2 && o.start_solve != 3 && o.N >= 40 && !overlap_linearise &&
                          o.forward_sweep != 2 && o.forward_split != -1;
    if (o.forward_split == 1 && !split_ok) { delete s; return CFNMPC_EINVAL; }
    // forward sweep on the stored blocks (row groups) below 6 S instances where the split matrix-free sweep takes over from there;
    // where it cannot run (short horizons, no dense structure, ...) the unsplit matrix-free sweep wins from 8 S on only
    // (an explicitly fused start solve stores no stage blocks: the automatic choice then stays with the matrix-free sweep)
    {
        const bool auto_rg = split_ok ? pick.forward_rg : (long)batch < 8L * simds;
        P.forward_rg = o.forward_sweep == 2 || (o.forward_sweep == 0 && auto_rg && o.start_solve != 2 && !(o.forward_split == 1)) ? 1 : 0;
    }
    if (P.as_dense) {   // side stream of the rows with long heads (launch_qp_ipm); without it the two kernels simply run one after the other
        hipStream_t side = nullptr;
        hipEvent_t ef = nullptr, ej = nullptr;
        if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&ef, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ej, hipEventDisableTiming) == hipSuccess) {
            P.as_side = side; P.as_fork = ef; P.as_join = ej;
            hipStream_t side2 = nullptr;
            hipEvent_t ej2 = nullptr;
            if (hipStreamCreateWithFlags(&side2, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&ej2, hipEventDisableTiming) == hipSuccess) {
                P.as_side2 = side2; P.as_join2 = ej2;
            } else {
                if (side2) (void)hipStreamDestroy(side2);
                (void)hipGetLastError();
            }
        } else {
            if (side) (void)hipStreamDestroy(side);
            if (ef) (void)hipEventDestroy(ef);
            (void)hipGetLastError();
        }
    }
    // start solve: the fused kernel replaces k_linearise + k_factor where nothing but the constrained instances' QP kernels
    // reads the stage blocks afterwards (matrix-free forward sweep, monolithic active-set kernel, no partial condensing,
    // no overlapped preparation); per-stage boxes (cfnmpc_set_box_stages) switch a solver back at launch time
    {
        const bool can_fuse = !cond_N2 && !overlap_linearise && !P.forward_rg && P.as_passes == 0;
        if (o.start_solve == 2 && !can_fuse) { cfnmpc_free(s); return CFNMPC_EINVAL; }
        if (o.start_solve == 3 && (cond_N2 || overlap_linearise)) { cfnmpc_free(s); return CFNMPC_EINVAL; }
        P.fused = o.start_solve == 2 ? 1 : (o.start_solve == 3 ? 2 : 0);
        P.clist_chunks = o.N >= 10 ? 10 : o.N;
    }
    // split forward sweep: lives in the dense structure (two side streams), matrix-free sweep, horizons that reach well behind H
    // (without the second side stream -- its creation failed -- the sweep stays in one launch)
    P.fwd_split = (split_ok && P.as_side2 && !P.forward_rg && P.fused == 0 && (o.forward_split == 1 || pick.forward_split)) ? 24 : 0;
    P.cond_N2 = cond_N2;
    P.ab16 = 1;                // home A, B, b grouped by 16 blocks (cfnmpc_rg.hpp: abidx)
    P.v4b = cond_N2 ? 0 : 1;   // home 4-vectors wave-blocked (the condensed kernels index theirs instance-major)
    P.cond_M = cond_N2 ? o.N / cond_N2 : 0;
    P.cond_rem = cond_N2 ? o.N % cond_N2 : 0;
    // one spare workspace block (index P.NW) parks the idle rows of compacted interior-point waves
    const size_t NW = P.NW + 1, N = P.N;
    int rc = CFNMPC_OK;
#define ALLOC(field, cnt) if (rc == CFNMPC_OK) rc = dev_alloc(s, &P.field, (size_t)(cnt))
    ALLOC(xit, NW * (N + 1) * cfn::SZ_V13); ALLOC(uit, NW * 4 * N * 4);
    ALLOC(xitn, NW * (N + 1) * cfn::SZ_V13); ALLOC(uitn, NW * 4 * N * 4); ALLOC(x0, NW * cfn::SZ_V13);
    ALLOC(yref, NW * N * cfn::SZ_Y); ALLOC(yref_e, NW * cfn::SZ_V13);
    // (the linearisation's home fields are laid out and written in groups of 16 blocks, P.ab16)
    const size_t NW16 = ((size_t)P.NW / 16 + 1) * 16;   // whole groups of 16 blocks, the spare block NW included (cfnmpc_rg.hpp: abidx)
    ALLOC(AR, NW16 * N * cfn::SZ_A); ALLOC(BR, NW16 * N * cfn::SZ_B); ALLOC(b, NW16 * N * cfn::SZ_V13);
    ALLOC(KR, NW * N * cfn::SZ_K); ALLOC(Sinv, NW * N * cfn::SZ_S);
    ALLOC(d, NW * 4 * N * 4); ALLOC(Pchk, NW * cfn::N_CHK * cfn::SZ_PP);
    ALLOC(v, NW * 4 * N * 4); ALLOC(tl, NW * 4 * N * 4); ALLOC(tu, NW * 4 * N * 4); ALLOC(ll, NW * 4 * N * 4);
    ALLOC(lu, NW * 4 * N * 4); ALLOC(rg, NW * 4 * N * 4); ALLOC(dva, NW * 4 * N * 4); ALLOC(dvc, NW * 4 * N * 4);
    ALLOC(Rh, NW * 4 * N * 4); ALLOC(g, NW * 4 * N * 4);
    ALLOC(cAR, NW * N * cfn::SZ_A); ALLOC(cBR, NW * N * cfn::SZ_B); ALLOC(cKR, NW * N * cfn::SZ_K);
    ALLOC(cSinv, NW * N * cfn::SZ_S); ALLOC(cd, NW * 4 * N * 4); ALLOC(cPchk, NW * cfn::N_CHK * cfn::SZ_P);
    ALLOC(cv, NW * 4 * N * 4); ALLOC(cuit, NW * 4 * N * 4); ALLOC(cGR, NW * N * cfn::SZ_K);
    ALLOC(cS, NW * N * cfn::SZ_S4); ALLOC(crho, NW * 4 * N * 4);
    if (P.fused == 1) ALLOC(cbv, NW * N * cfn::SZ_V13);
    ALLOC(cPs, NW * 32 * cfn::SZ_PA);
    ALLOC(status, NW * 4); ALLOC(iters, NW * 4); ALLOC(head, NW * 4); ALLOC(res, NW * 4); ALLOC(viol, NW * 4);
    ALLOC(ilist, NW * 4); ALLOC(ilist2, NW * 4); ALLOC(nipm, 64);
    ALLOC(blkcnt, ((size_t)(batch + 63) / 64) * cfn::BIN_STRIDE); ALLOC(rank, NW * 4); ALLOC(done, NW * 4);
    ALLOC(ascnt, 32); ALLOC(askst, NW * 4); ALLOC(asst, NW * 4); ALLOC(asok, NW * 4);
    ALLOC(czdx, NW * 4 * (N + 1) * 13);
    if (P.as_warm) { ALLOC(wcls, NW * 4 * N * 4); ALLOC(wvalid, NW * 4); }
    if (P.fwd_split) { ALLOC(fs_dx, ((size_t)(batch + 63) / 64) * 13 * 64); ALLOC(fs_st, ((size_t)(batch + 63) / 64) * 4 * 64); }
    if (P.as_passes != 0) ALLOC(aslist, (size_t)3 * 7 * NW * 4);
    if (cond_N2) ALLOC(cb, NW * 4 * (size_t)cond_N2 * cfn::cb_size(cfn::cond_mmax(P)));
    if (s->overlap) {
        if (rc == CFNMPC_OK) rc = dev_alloc(s, &s->AR2, NW16 * N * cfn::SZ_A);
        if (rc == CFNMPC_OK) rc = dev_alloc(s, &s->BR2, NW16 * N * cfn::SZ_B);
        if (rc == CFNMPC_OK) rc = dev_alloc(s, &s->b2, NW16 * N * cfn::SZ_V13);
        int lo = 0, hi = 0;  // lo = least priority (numerically greatest)
        if (rc == CFNMPC_OK && (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess ||
                                hipStreamCreateWithPriority(&s->aux, hipStreamNonBlocking, lo) != hipSuccess ||
                                hipEventCreateWithFlags(&s->ev_start, hipEventDisableTiming) != hipSuccess ||
                                hipEventCreateWithFlags(&s->ev_aux, hipEventDisableTiming) != hipSuccess))
            rc = CFNMPC_EHIP;
    }
#undef ALLOC
    s->stage_doubles = (size_t)batch * (N + 1) * 17;
    if (rc == CFNMPC_OK) rc = dev_alloc(s, &s->stage_buf, s->stage_doubles);
    if (rc != CFNMPC_OK) { cfnmpc_free(s); return rc; }
    // default iterate = what acados_create() leaves behind
    cfn::launch_init_iterate(P, CFNMPC_INIT_ACADOS, nullptr);
    if (hipDeviceSynchronize() != hipSuccess) { cfnmpc_free(s); return CFNMPC_EHIP; }
    *out = s;
    return CFNMPC_OK;
}

int cfnmpc_free(cfnmpc_solver* s) {
    if (!s) return CFNMPC_EINVAL;
    DeviceGuard dg(s);
    if (s->aux) { (void)hipStreamSynchronize(s->aux); (void)hipStreamDestroy(s->aux); }
    if (s->P.as_side2) {
        (void)hipStreamSynchronize((hipStream_t)s->P.as_side2);
        (void)hipStreamDestroy((hipStream_t)s->P.as_side2);
        (void)hipEventDestroy((hipEvent_t)s->P.as_join2);
    }
    if (s->P.as_side) {
        (void)hipStreamSynchronize((hipStream_t)s->P.as_side);
        (void)hipStreamDestroy((hipStream_t)s->P.as_side);
        (void)hipEventDestroy((hipEvent_t)s->P.as_fork);
        (void)hipEventDestroy((hipEvent_t)s->P.as_join);
    }
    for (int p = 0; p < 2; p++) {
        if (s->glaunched[p]) { (void)hipEventSynchronize(s->glaunched[p]); (void)hipEventDestroy(s->glaunched[p]); }
        if (s->gexec[p]) (void)hipGraphExecDestroy(s->gexec[p]);
    }
    if (s->cap) (void)hipStreamDestroy(s->cap);
    if (s->ev_start) (void)hipEventDestroy(s->ev_start);
    if (s->ev_aux) (void)hipEventDestroy(s->ev_aux);
    for (void* p : s->allocs) (void)hipFree(p);
    if (s->h_io) (void)hipHostFree(s->h_io);
    for (hipEvent_t e : s->ev) (void)hipEventDestroy(e);
    delete s;
    return CFNMPC_OK;
}

int cfnmpc_batch(const cfnmpc_solver* s) { return s ? s->P.B : CFNMPC_EINVAL; }
int cfnmpc_horizon(const cfnmpc_solver* s) { return s ? s->P.N : CFNMPC_EINVAL; }
unsigned long long cfnmpc_workspace_bytes(const cfnmpc_solver* s) { return s ? s->bytes : 0ull; }

int cfnmpc_set_x0(cfnmpc_solver* s, const double* x0, int on_device, void* stream) {
    if (!s || !x0) return CFNMPC_EINVAL;
    DeviceGuard dg(s);
    return put_field(s, x0, on_device, 1, 13, 1, s->P.x0, (hipStream_t)stream);
}

int cfnmpc_set_yref(cfnmpc_solver* s, const double* yref, const double* yref_e, int on_device, void* stream) {
    if (!s || !yref || !yref_e) return CFNMPC_EINVAL;
    DeviceGuard dg(s);
    int rc = put_field(s, yref, on_device, s->P.N, 17, 1, s->P.yref, (hipStream_t)stream);
    if (rc != CFNMPC_OK) return rc;
    return put_field(s, yref_e, on_device, 1, 13, 1, s->P.yref_e, (hipStream_t)stream);
}

int cfnmpc_set_yref_windows(cfnmpc_solver* s, const double* traj, int n_rows, int* mode, int* iter,
                            const double* des_xyz, double uss, void* stream) {
    if (!s || !mode || !iter || !des_xyz) return CFNMPC_EINVAL;
    DeviceGuard dg(s);
    if (n_rows > 0 && (!traj || n_rows < s->P.N + 1)) return CFNMPC_EINVAL;
    if (n_rows <= 0 && traj) return CFNMPC_EINVAL;
    cfn::launch_windows(s->P, traj, n_rows > 0 ? n_rows : 0, mode, iter, des_xyz, uss, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return CFNMPC_OK;
}

// kernel arguments changed (weights, box): the captured steps hold the old ones
static void invalidate_graphs(cfnmpc_solver* s) { s->gvalid[0] = s->gvalid[1] = false; }

int cfnmpc_set_weights(cfnmpc_solver* s, const double* W, const double* WN) {
    if (!s || (!W && !WN)) return CFNMPC_EINVAL;
    if (!weights_ok(W, WN)) return CFNMPC_EINVAL;   // validated as a whole before anything is copied
    if (W) for (int i = 0; i < 17; i++) s->P.W[i] = W[i];
    if (WN) for (int i = 0; i < 13; i++) s->P.WN[i] = WN[i];
    invalidate_graphs(s);
    return CFNMPC_OK;  // kernel arguments: take effect at the next cfnmpc_solve
}

int cfnmpc_set_box(cfnmpc_solver* s, double u_min, double u_max) {
    if (!s || !(u_max > u_min)) return CFNMPC_EINVAL;
    s->P.u_min = u_min;   // kernel arguments: take effect at the next cfnmpc_solve
    s->P.u_max = u_max;
    invalidate_graphs(s);
    return CFNMPC_OK;
}

int cfnmpc_set_box_stages(cfnmpc_solver* s, const double* lb, const double* ub, int on_device, void* stream) {
    if (!s || ((lb == nullptr) != (ub == nullptr))) return CFNMPC_EINVAL;
    DeviceGuard dg(s);
    cfn::Params& P = s->P;
    if (!lb) {                       // back to the scalar box of cfnmpc_set_box
        if (P.lbs) { s->lbs_keep = P.lbs; s->ubs_keep = P.ubs; }
        P.lbs = P.ubs = nullptr;
        invalidate_graphs(s);
        return CFNMPC_OK;
    }
    if (P.cond_N2) return CFNMPC_EINVAL;   // the condensed path has no per-stage boxes
    const size_t n = (size_t)P.B * P.N * 4;
    // host arrays are validated here (NaN fails too; lb = ub pins the input); DEVICE arrays are taken as they are -- a caller
    // that hands over device pointers is responsible for lb <= ub (an inverted box ends in status 4 for that vehicle)
    if (is_host(on_device))
        for (size_t i = 0; i < n; i++) if (!(lb[i] <= ub[i])) return CFNMPC_EINVAL;
    if (!s->lbs_keep || !P.clbs || !P.cubs) {
        // allocated and initialised as a whole before any pointer is committed: a failure half-way (ENOMEM, a failed copy)
        // leaves the solver on the scalar box with nothing dangling (the blocks stay owned by s->allocs until cfnmpc_free)
        const size_t cnt = ((size_t)P.NW + 1) * 4 * P.N * 4;
        // (blocks a failed earlier attempt did get are kept in s->box_blk and reused: a retry allocates only what is missing)
        int rc = CFNMPC_OK;
        for (int q = 0; q < 4 && rc == CFNMPC_OK; q++)
            if (!s->box_blk[q]) rc = dev_alloc(s, &s->box_blk[q], cnt);
        if (rc != CFNMPC_OK) return rc;
        double *lk = s->box_blk[0], *uk = s->box_blk[1], *cl = s->box_blk[2], *cu = s->box_blk[3];
        // rows of the spare block (parked rows of compacted waves): a wide finite box
        std::vector<double> lo(4 * (size_t)P.N * 4, -1e30), hi(4 * (size_t)P.N * 4, 1e30);
        HIP_TRY(hipMemcpy(lk + (size_t)P.NW * 4 * P.N * 4, lo.data(), lo.size() * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(uk + (size_t)P.NW * 4 * P.N * 4, hi.data(), hi.size() * 8, hipMemcpyHostToDevice));
        s->lbs_keep = lk; s->ubs_keep = uk; P.clbs = cl; P.cubs = cu;
    }
    // the caller's AoS order [inst][stage][4] -> the layout of the home 4-vectors (Params.v4b).  Both arrays are STAGED first
    // (in the compact box buffers: scratch of the QP kernels, rewritten by every solve that uses them) and the home pair is
    // overwritten only after both puts succeeded -- a failure leaves the previous lb AND ub in force, never a mixed box.
    hipStream_t st = (hipStream_t)stream;
    int rcp = put_field(s, lb, on_device, P.N, 4, 0, P.clbs, st);
    if (rcp == CFNMPC_OK) rcp = put_field(s, ub, on_device, P.N, 4, 0, P.cubs, st);
    if (rcp != CFNMPC_OK) return rcp;
    const size_t bytes = (size_t)P.NW * 4 * P.N * 4 * sizeof(double);   // (the spare block behind it keeps its wide box)
    HIP_TRY(hipMemcpyAsync(s->lbs_keep, P.clbs, bytes, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->ubs_keep, P.cubs, bytes, hipMemcpyDeviceToDevice, st));
    if (on_device == CFNMPC_ON_HOST) HIP_TRY(hipStreamSynchronize(st));   // synchronous form: complete when the call returns
    P.lbs = s->lbs_keep; P.ubs = s->ubs_keep;
    invalidate_graphs(s);
    return CFNMPC_OK;
}

int cfnmpc_get_cmd(cfnmpc_solver* s, double* cmd_vel, int* motvel, int on_device, void* stream) {
    if (!s || !cmd_vel) return CFNMPC_EINVAL;
    DeviceGuard dg(s);
    hipStream_t st = (hipStream_t)stream;
    const size_t B = s->P.B;
    if (!is_host(on_device)) {
        cfn::launch_postproc(s->P, cmd_vel, motvel, st);
        HIP_TRY(hipGetLastError());
        return CFNMPC_OK;
    }
    // host pointers: through the staging buffer ([B][4] doubles, then [B][4] ints)
    if (B * 6 > s->stage_doubles) return CFNMPC_EINVAL;
    double* dc = s->stage_buf;
    int* dm = (int*)(s->stage_buf + B * 4);
    cfn::launch_postproc(s->P, dc, dm, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(cmd_vel, dc, B * 4 * sizeof(double), hipMemcpyDeviceToHost, st));
    if (motvel) HIP_TRY(hipMemcpyAsync(motvel, dm, B * 4 * sizeof(int), hipMemcpyDeviceToHost, st));
    if (on_device == CFNMPC_ON_HOST) HIP_TRY(hipStreamSynchronize(st));
    return CFNMPC_OK;
}

int cfnmpc_init_iterate(cfnmpc_solver* s, int mode, void* stream) {
    if (!s || (mode != CFNMPC_INIT_ACADOS && mode != CFNMPC_INIT_HOVER)) return CFNMPC_EINVAL;
    DeviceGuard dg(s);
    cfn::launch_init_iterate(s->P, mode, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    if (s->P.as_warm) HIP_TRY(hipMemsetAsync(s->P.wvalid, 0, sizeof(int) * (size_t)s->P.B, (hipStream_t)stream));   // a new iterate: no set to start from
    s->lin_valid = false;
    return CFNMPC_OK;
}

int cfnmpc_set_iterate(cfnmpc_solver* s, const double* x, const double* u, int on_device, void* stream) {
    if (!s || !x || !u) return CFNMPC_EINVAL;
    DeviceGuard dg(s);
    s->lin_valid = false;
    if (s->P.as_warm) HIP_TRY(hipMemsetAsync(s->P.wvalid, 0, sizeof(int) * (size_t)s->P.B, (hipStream_t)stream));
    int rc = put_field(s, x, on_device, s->P.N + 1, 13, 1, s->P.xit, (hipStream_t)stream);
    if (rc != CFNMPC_OK) return rc;
    return put_field(s, u, on_device, s->P.N, 4, 0, s->P.uit, (hipStream_t)stream);
}

int cfnmpc_get_iterate(cfnmpc_solver* s, double* x, double* u, int on_device, void* stream) {
    if (!s || !x || !u) return CFNMPC_EINVAL;
    DeviceGuard dg(s);
    int rc = get_field(s, x, on_device, s->P.N + 1, 13, 1, 0, s->P.N + 1, s->P.xit, (hipStream_t)stream);
    if (rc != CFNMPC_OK) return rc;
    return get_field(s, u, on_device, s->P.N, 4, 0, 0, s->P.N, s->P.uit, (hipStream_t)stream);
}

int cfnmpc_solve(cfnmpc_solver* s, int n_rti, void* stream) {
    if (!s || n_rti < 1) return CFNMPC_EINVAL;
    DeviceGuard dg(s);
    hipStream_t st = (hipStream_t)stream;
    for (int it = 0; it < n_rti; it++) {
        hipEvent_t* e = nullptr;
        if (s->profiling && s->ev_used < EV_PER_STEP * MAX_PROFILED_STEPS) {   // bounded: later steps go untimed
            while (s->ev.size() < s->ev_used + EV_PER_STEP) {
                hipEvent_t ne;
                HIP_TRY(hipEventCreate(&ne));
                s->ev.push_back(ne);
            }
            e = &s->ev[s->ev_used];
            s->ev_used += EV_PER_STEP;
        }
        if (s->reinit_failed) { cfn::launch_reinit_failed(s->P, st); s->lin_valid = false; }
        if (!s->overlap && s->use_graph && !e) {
            // the step's launches replayed from a captured graph (one per parity of the iterate buffers)
            const int p = s->parity;
            if (!s->gvalid[p]) {
                if (!s->cap) HIP_TRY(hipStreamCreateWithFlags(&s->cap, hipStreamNonBlocking));
                if (!s->glaunched[p]) HIP_TRY(hipEventCreateWithFlags(&s->glaunched[p], hipEventDisableTiming));
                if (s->gexec[p]) {   // a launch of the old exec may still be running (cfnmpc_solve is asynchronous)
                    HIP_TRY(hipEventSynchronize(s->glaunched[p]));
                    (void)hipGraphExecDestroy(s->gexec[p]);
                    s->gexec[p] = nullptr;
                }
                hipGraph_t g = nullptr;
                bool ok = hipStreamBeginCapture(s->cap, hipStreamCaptureModeThreadLocal) == hipSuccess;
                if (ok) {
                    if (s->P.fused != 1 || s->P.lbs) cfn::launch_linearise(s->P, s->chunks_all, s->cap);
                    if (s->P.cond_N2) cfn::launch_qp_cond(s->P, s->cap);
                    else cfn::launch_qp(s->P, s->cap);
                    ok = hipStreamEndCapture(s->cap, &g) == hipSuccess && g != nullptr;   // (always ends the capture)
                }
                if (ok) ok = hipGraphInstantiate(&s->gexec[p], g, nullptr, nullptr, 0) == hipSuccess;
                if (g) (void)hipGraphDestroy(g);
                if (!ok) {
                    // capture / instantiation failed: drop the capture stream (it may be left in an invalid capture
                    // state) and fall back to individual launches for good
                    (void)hipGetLastError();
                    (void)hipStreamDestroy(s->cap);
                    s->cap = nullptr;
                    s->gexec[p] = nullptr;
                    s->use_graph = 0;
                    std::fprintf(stderr, "cfnmpc: step_graph capture failed, launching the step's kernels individually\n");
                    it--;          // redo this step on the plain path
                    continue;
                }
                s->gvalid[p] = true;
            }
            HIP_TRY(hipGraphLaunch(s->gexec[p], st));
            HIP_TRY(hipEventRecord(s->glaunched[p], st));
            std::swap(s->P.xit, s->P.xitn);
            std::swap(s->P.uit, s->P.uitn);
            s->parity ^= 1;
            s->lin_valid = false;
            continue;
        }
        if (!s->overlap) {
            // linearise -> QP, everything on the caller's stream
            if (e) HIP_TRY(hipEventRecord(e[0], st));
            if (s->P.fused != 1 || s->P.lbs) cfn::launch_linearise(s->P, s->chunks_all, st);   // (fused start solve: k_linfactor linearises)
            if (e) HIP_TRY(hipEventRecord(e[1], st));
            if (s->P.cond_N2) {
                cfn::launch_qp_cond(s->P, st);   // pcond -> condensed Riccati -> expand (-> interior point)
                if (e) for (int j = 2; j < 6; j++) HIP_TRY(hipEventRecord(e[j], st));   // (no per-kernel split on this path)
            } else {
                cfn::launch_qp(s->P, st, e ? e + 2 : nullptr);
            }
            if (e) HIP_TRY(hipEventRecord(e[6], st));
            std::swap(s->P.xit, s->P.xitn);   // the step's kernels wrote every instance's new iterate there
            std::swap(s->P.uit, s->P.uitn);
            s->parity ^= 1;
            s->lin_valid = false;   // the iterate moved
            continue;
        }
#ifdef CFN_DEV   // overlapped preparation: development builds only (CFNMPC_OVERLAP=1); s->overlap is 0 in the product
        // feedback phase on the linearisation prepared by the previous step ...
        if (!s->lin_valid) cfn::launch_linearise(s->P, s->chunks_all, st);
        if (e) HIP_TRY(hipEventRecord(e[0], st));
        cfn::launch_qp_start(s->P, st);
        HIP_TRY(hipEventRecord(s->ev_start, st));
        cfn::launch_qp_ipm(s->P, st);
        if (e) { for (int j = 1; j < 6; j++) HIP_TRY(hipEventRecord(e[j], st)); }   // (phases only on the overlapped path)
        // ... and preparation of the next step into the alternate set: an early pass over ALL
        // instances runs beside the interior-point kernel (the instances still inside it are
        // linearised around a stale iterate there and redone by the list pass afterwards)
        std::swap(s->P.xit, s->P.xitn);   // (host-side: kernel arguments are by value)
        std::swap(s->P.uit, s->P.uitn);
        cfn::Params Q = s->P;
        Q.AR = s->AR2; Q.BR = s->BR2; Q.b = s->b2;
        HIP_TRY(hipStreamWaitEvent(s->aux, s->ev_start, 0));
        cfn::launch_linearise(Q, s->chunks_all, s->aux);
        HIP_TRY(hipEventRecord(s->ev_aux, s->aux));
        HIP_TRY(hipStreamWaitEvent(st, s->ev_aux, 0));
        cfn::launch_linearise_list(Q, s->chunks_list, st);
        if (e) HIP_TRY(hipEventRecord(e[6], st));
        s->AR2 = s->P.AR; s->BR2 = s->P.BR; s->b2 = s->P.b;
        s->P.AR = Q.AR; s->P.BR = Q.BR; s->P.b = Q.b;
        s->lin_valid = true;
#endif
    }
    HIP_TRY(hipGetLastError());
    return CFNMPC_OK;
}

int cfnmpc_step_host(cfnmpc_solver* s, const double* x0, const double* yref, const double* yref_e, double* u,
                     double* x, int* status, int* qp_iter, double* res, void* stream) {
    if (!s || !x0 || !yref || !yref_e) return CFNMPC_EINVAL;
    DeviceGuard dg(s);
    hipStream_t st = (hipStream_t)stream;
    const cfn::Params& P = s->P;
    const size_t B = P.B, N = P.N;
    // layout of the I/O block (doubles): x0 | yref | yref_e || u | x | res | status, iters (as ints)
    const size_t n_x0 = B * 13, n_yr = B * N * 17, n_ye = B * 13, n_in = n_x0 + n_yr + n_ye;
    const size_t n_u = B * N * 4, n_x = B * (N + 1) * 13, n_res = B, n_int = B;   // 2 ints per double slot
    const size_t n_out = n_u + n_x + n_res + n_int, n_all = n_in + n_out;
    if (s->io_doubles < n_all) {
        if (s->h_io) (void)hipHostFree(s->h_io);
        s->h_io = nullptr;
        HIP_TRY(hipHostMalloc((void**)&s->h_io, n_all * sizeof(double), hipHostMallocDefault));
        int rc = dev_alloc(s, &s->d_io, n_all);
        if (rc != CFNMPC_OK) return rc;
        s->io_doubles = n_all;
    }
    double *h = s->h_io, *d = s->d_io;
    std::memcpy(h, x0, n_x0 * sizeof(double));
    std::memcpy(h + n_x0, yref, n_yr * sizeof(double));
    std::memcpy(h + n_x0 + n_yr, yref_e, n_ye * sizeof(double));
    HIP_TRY(hipMemcpyAsync(d, h, n_in * sizeof(double), hipMemcpyHostToDevice, st));
    cfn::launch_put(P.B, 1, 13, 1, d, P.x0, st);
    cfn::launch_put(P.B, P.N, 17, 1, d + n_x0, P.yref, st);
    cfn::launch_put(P.B, 1, 13, 1, d + n_x0 + n_yr, P.yref_e, st);
    int rc = cfnmpc_solve(s, 1, stream);
    if (rc != CFNMPC_OK) return rc;
    double* o = d + n_in;
    cfn::launch_get(P.B, P.N, 4, 0, 0, P.N, s->P.uit, o, st, P.v4b);
    cfn::launch_get(P.B, P.N + 1, 13, 1, 0, P.N + 1, s->P.xit, o + n_u, st);
    HIP_TRY(hipMemcpyAsync(o + n_u + n_x, s->P.res, B * sizeof(double), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(o + n_u + n_x + n_res, s->P.status, B * sizeof(int), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync((int*)(o + n_u + n_x + n_res) + B, s->P.iters, B * sizeof(int), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h + n_in, o, n_out * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const double* ho = h + n_in;
    if (u) std::memcpy(u, ho, n_u * sizeof(double));
    if (x) std::memcpy(x, ho + n_u, n_x * sizeof(double));
    if (res) std::memcpy(res, ho + n_u + n_x, B * sizeof(double));
    const int* hi = (const int*)(ho + n_u + n_x + n_res);
    if (status) std::memcpy(status, hi, B * sizeof(int));
    if (qp_iter) std::memcpy(qp_iter, hi + B, B * sizeof(int));
    return CFNMPC_OK;
}

int cfnmpc_get_u(cfnmpc_solver* s, int stage, double* u, int on_device, void* stream) {
    if (!s || !u || stage < 0 || stage >= s->P.N) return CFNMPC_EINVAL;
    DeviceGuard dg(s);
    return get_field(s, u, on_device, 1, 4, 0, stage, s->P.N, s->P.uit, (hipStream_t)stream);
}

int cfnmpc_get_x(cfnmpc_solver* s, int stage, double* x, int on_device, void* stream) {
    if (!s || !x || stage < 0 || stage > s->P.N) return CFNMPC_EINVAL;
    DeviceGuard dg(s);
    return get_field(s, x, on_device, 1, 13, 1, stage, s->P.N + 1, s->P.xit, (hipStream_t)stream);
}

int cfnmpc_get_stats(cfnmpc_solver* s, int* status, int* qp_iter, double* res, int on_device, void* stream) {
    if (!s) return CFNMPC_EINVAL;
    DeviceGuard dg(s);
    hipStream_t st = (hipStream_t)stream;
    const hipMemcpyKind kind = !is_host(on_device) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    const size_t B = s->P.B;
    if (status) HIP_TRY(hipMemcpyAsync(status, s->P.status, B * sizeof(int), kind, st));
    if (qp_iter) HIP_TRY(hipMemcpyAsync(qp_iter, s->P.iters, B * sizeof(int), kind, st));
    if (res) HIP_TRY(hipMemcpyAsync(res, s->P.res, B * sizeof(double), kind, st));
    if (on_device == CFNMPC_ON_HOST) HIP_TRY(hipStreamSynchronize(st));
    return CFNMPC_OK;
}

int cfnmpc_sim(int batch, const double* x, const double* u, double T, int steps, double* xn, int on_device,
               void* stream) {
    if (batch <= 0 || !x || !u || !xn || steps < 1 || !(T > 0)) return CFNMPC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (!is_host(on_device)) {
        cfn::launch_sim(batch, x, u, T, steps, xn, st);
        HIP_TRY(hipGetLastError());
        return CFNMPC_OK;
    }
    // host pointers: device scratch cached per device and grown on demand (the estimator calls this
    // at 66 Hz with batch 1, acados_estimator.cpp:589 -- no allocation per call); one caller at a time
    static std::mutex mtx[64];             // one caller at a time PER DEVICE (predictors on different GPUs run concurrently)
    static double* scratch[64] = {nullptr};
    static size_t cap[64] = {0};
    int devi = 0;
    HIP_TRY(hipGetDevice(&devi));
    if (devi < 0 || devi >= 64) return CFNMPC_EHIP;
    std::lock_guard<std::mutex> lock(mtx[devi]);
    const size_t B = batch, need = B * 30;
    if (cap[devi] < need) {
        if (scratch[devi]) (void)hipFree(scratch[devi]);
        scratch[devi] = nullptr; cap[devi] = 0;
        if (hipMalloc((void**)&scratch[devi], need * 8) != hipSuccess) return CFNMPC_ENOMEM;
        cap[devi] = need;
    }
    double *dx = scratch[devi], *du = dx + B * 13, *dn = du + B * 4;
    HIP_TRY(hipMemcpyAsync(dx, x, B * 13 * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(du, u, B * 4 * 8, hipMemcpyHostToDevice, st));
    cfn::launch_sim(batch, dx, du, T, steps, dn, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(xn, dn, B * 13 * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return CFNMPC_OK;
}

int cfnmpc_estimate(int batch, const double* meas, double* filt, const double* u, double dt, int use_lpf, double delay,
                    int steps, double* x_est, double* x_pred, void* stream) {
    if (batch <= 0 || !meas || !filt || !u || !x_est || !x_pred || steps < 1 || !(delay > 0) || !(dt > 0))
        return CFNMPC_EINVAL;
    cfn::launch_estimate(batch, meas, filt, u, dt, use_lpf ? 1 : 0, delay, steps, x_est, x_pred, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return CFNMPC_OK;
}

int cfnmpc_set_profiling(cfnmpc_solver* s, int enable) {
    if (!s) return CFNMPC_EINVAL;
    s->profiling = enable ? 1 : 0;
    s->ev_used = 0;
    return CFNMPC_OK;
}

// per timed step: ms_steps [max_steps][6] (the six groups of cfnmpc_get_profile_kernels), *n_steps = steps written (<= max_steps;
// later timed steps are dropped); resets the count like cfnmpc_get_profile_kernels
int cfnmpc_get_profile_steps(cfnmpc_solver* s, double* ms_steps, int max_steps, int* n_steps) {
    if (!s || !ms_steps || max_steps < 0 || !n_steps) return CFNMPC_EINVAL;
    DeviceGuard dg(s);
    size_t n = s->ev_used / EV_PER_STEP;
    if (n > (size_t)max_steps) n = (size_t)max_steps;
    for (size_t i = 0; i < n; i++) {
        hipEvent_t* e = &s->ev[EV_PER_STEP * i];
        HIP_TRY(hipEventSynchronize(e[EV_PER_STEP - 1]));
        for (size_t j = 0; j + 1 < EV_PER_STEP; j++) {
            float t = 0.f;
            HIP_TRY(hipEventElapsedTime(&t, e[j], e[j + 1]));
            ms_steps[i * (EV_PER_STEP - 1) + j] = t;
        }
    }
    *n_steps = (int)n;
    s->ev_used = 0;
    return CFNMPC_OK;
}

int cfnmpc_get_profile_kernels(cfnmpc_solver* s, double* ms, int* n_steps) {
    if (!s || !ms || !n_steps) return CFNMPC_EINVAL;
    const size_t n = s->ev_used / EV_PER_STEP;
    std::vector<double> per(n * (EV_PER_STEP - 1) + 1);
    int got = 0;
    const int rc = cfnmpc_get_profile_steps(s, per.data(), (int)n, &got);
    if (rc != CFNMPC_OK) return rc;
    for (size_t j = 0; j + 1 < EV_PER_STEP; j++) {
        double acc = 0.0;
        for (int i = 0; i < got; i++) acc += per[(size_t)i * (EV_PER_STEP - 1) + j];
        ms[j] = got ? acc / got : 0.0;
    }
    *n_steps = got;
    return CFNMPC_OK;
}

int cfnmpc_get_profile(cfnmpc_solver* s, double* ms_linearise, double* ms_qp, int* n_steps) {
    if (!s || !ms_linearise || !ms_qp || !n_steps) return CFNMPC_EINVAL;
    double ms[EV_PER_STEP - 1];
    const int rc = cfnmpc_get_profile_kernels(s, ms, n_steps);
    if (rc != CFNMPC_OK) return rc;
    const double a = ms[0], b = ms[1] + ms[2] + ms[3] + ms[4] + ms[5];
    // overlap: the QP phase comes first (events 0 -> 1..5), then the exposed linearisation (5 -> 6)
    *ms_linearise = s->overlap ? ms[5] : a;
    *ms_qp = s->overlap ? a : b;
    return CFNMPC_OK;
}

#ifdef CFN_PROF
extern "C++" { namespace cfn { void debug_prof_read(unsigned long long* out, int reset);
                               float debug_bench_sweep(const Params& P, int waves, int head, int reps, int which); } }
float cfnmpc_debug_bench_sweep(cfnmpc_solver* s, int waves, int head, int reps, int which) {
    return cfn::debug_bench_sweep(s->P, waves, head, reps, which);
}
extern "C++" { namespace cfn { void debug_dprof_read(unsigned long long* out, int reset); } }
int cfnmpc_debug_dprof(unsigned long long* out, int reset) { (void)hipDeviceSynchronize(); cfn::debug_dprof_read(out, reset); return 0; }
int cfnmpc_debug_prof(unsigned long long* out, int reset) { (void)hipDeviceSynchronize(); cfn::debug_prof_read(out, reset); return 0; }
#endif

#ifdef CFN_DEV
// EXPERIMENT: linearisation + backward factorisation of the start solve, either as the two
// kernels of the product (chunk = 0) or alternating in chunks of `chunk` stages going backward -- linearise stages
// [k, k + chunk), then factorise them (cost-to-go parked between the launches) -- so that the stage blocks might be
// consumed from the L2 / MALL.  `reps` repetitions; *ms = average duration of one pair (HIP events).  Same numbers
// either way (bitwise: the same arithmetic).  Leaves the solver ready for cfnmpc_solve.
int cfnmpc_debug_chunked_pair(cfnmpc_solver* s, int chunk, int reps, double* ms, void* stream) {
    if (!s || chunk < 0 || reps < 1 || !ms) return CFNMPC_EINVAL;
    DeviceGuard dg(s);
    hipStream_t st = (hipStream_t)stream;
    cfn::Params P = s->P;
    if (chunk > 0 && !P.Ppark) {
        int rc = dev_alloc(s, &s->P.Ppark, ((size_t)P.NW + 1) * 13 * 64);
        if (rc != CFNMPC_OK) return rc;
        P.Ppark = s->P.Ppark;
    }
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, st));
    for (int r = 0; r < reps; r++) {
        if (chunk == 0) {
            cfn::launch_linearise(P, s->chunks_all, st);
            cfn::launch_factor_only(P, st);
        } else {
            for (int hi = P.N; hi > 0; hi -= chunk) {
                const int lo = hi - chunk > 0 ? hi - chunk : 0;
                cfn::Params Q = P;
                Q.lin_k0 = lo; Q.lin_k1 = hi; Q.fk_lo = lo; Q.fk_hi = hi;
                // the chunk's intervals spread over as many workgroups as the whole horizon gets normally
                int c = s->chunks_all < hi - lo ? s->chunks_all : hi - lo;
                cfn::launch_linearise(Q, c < 1 ? 1 : c, st);
                cfn::launch_factor_chunk(Q, st);
            }
        }
    }
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipEventSynchronize(e1));
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms = t / reps;
    HIP_TRY(hipGetLastError());
    return CFNMPC_OK;
}

// sums of the start solve's gains, feed-forward terms and Riccati checkpoints (host side, exact order: equal sums for
// bitwise-equal arrays) -- checker of the chunked experiment
int cfnmpc_debug_checksum(cfnmpc_solver* s, double* out3) {
    if (!s || !out3) return CFNMPC_EINVAL;
    DeviceGuard dg(s);
    const cfn::Params& P = s->P;
    HIP_TRY(hipDeviceSynchronize());
    const size_t n[3] = {(size_t)P.NW * P.N * cfn::SZ_K, (size_t)P.NW * 4 * P.N * 4, (size_t)P.NW * cfn::N_CHK * cfn::SZ_PP};
    const double* src[3] = {P.KR, P.d, P.Pchk};
    for (int f = 0; f < 3; f++) {
        std::vector<double> h(n[f]);
        HIP_TRY(hipMemcpy(h.data(), src[f], n[f] * 8, hipMemcpyDeviceToHost));
        double acc = 0.0;
        for (size_t i = 0; i < n[f]; i++) acc += h[i] * (double)(1 + (i % 7));
        out3[f] = acc;
    }
    return CFNMPC_OK;
}
#endif   // CFN_DEV

// Start solve, backward half only, for parity tests and timing: mode 1 = k_linearise + k_factor, mode 2 = k_linfactor;
// `reps` repetitions timed with HIP events on `stream` (*ms = average per repetition; may be NULL).
int cfnmpc_debug_start_factor(cfnmpc_solver* s, int mode, int reps, double* ms, void* stream) {
    if (!s || (mode != 1 && mode != 2) || reps < 1) return CFNMPC_EINVAL;
    // k_factor / k_linfactor address the home 4-vectors in the wave-blocked layout (Params.v4b): a partial-condensing solver
    // keeps them instance-major and never runs these kernels -- refuse instead of reading and writing in the wrong layout
    if (s->P.cond_N2 || !s->P.v4b) return CFNMPC_EINVAL;
    DeviceGuard dg(s);
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess) return CFNMPC_EHIP;
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return CFNMPC_EHIP; }
    bool ok = hipEventRecord(e0, st) == hipSuccess;
    for (int r = 0; ok && r < reps; r++) {
        if (mode == 1) {
            cfn::launch_linearise(s->P, s->chunks_all, st);
            cfn::launch_factor_only(s->P, st);
        } else {
            cfn::launch_linfactor(s->P, st);
        }
    }
    float t = 0.f;
    ok = ok && hipEventRecord(e1, st) == hipSuccess && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&t, e0, e1) == hipSuccess;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (!ok) return CFNMPC_EHIP;
    if (ms) *ms = (double)t / reps;
    HIP_TRY(hipGetLastError());
    s->lin_valid = mode == 1;
    return CFNMPC_OK;
}

// What the start solve's backward sweep leaves behind, decoded into dense arrays in the EXTERNAL state order (host
// pointers, any may be NULL): gains K [B][N][4][13], feed-forward d [B][N][4], cost-to-go checkpoints Pchk [B][6][13][13]
// (stages 4, 8, 12, 16, 24, 32; only those below N are written by the kernels), status [B].
int cfnmpc_debug_get_factor(cfnmpc_solver* s, double* K, double* d, double* Pchk, int* status) {
    if (!s || s->P.cond_N2 || !s->P.v4b) return CFNMPC_EINVAL;   // (as cfnmpc_debug_start_factor)
    DeviceGuard dg(s);
    const cfn::Params& P = s->P;
    const size_t NW = P.NW, N = P.N, B = P.B;
    HIP_TRY(hipDeviceSynchronize());
    if (K) {
        std::vector<double> h(NW * N * cfn::SZ_K);
        HIP_TRY(hipMemcpy(h.data(), P.KR, h.size() * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < B; i++)
            for (size_t k = 0; k < N; k++) {
                const double* kb = h.data() + ((i / 4) * N + k) * cfn::SZ_K;
                for (int l = 0; l < 13; l++)
                    for (int a = 0; a < 4; a++) K[((i * N + k) * 4 + a) * 13 + cfn::ext_of(l)] = kb[(l * 4 + (i % 4)) * 4 + a];
            }
    }
    if (d) {
        std::vector<double> h(NW * 4 * N * 4);
        HIP_TRY(hipMemcpy(h.data(), P.d, h.size() * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < B; i++)
            for (size_t k = 0; k < N; k++)
                for (int a = 0; a < 4; a++)
                    d[(i * N + k) * 4 + a] = h[P.v4b ? (((i / 4) * N + k) * 4 + i % 4) * 4 + a : (i * N + k) * 4 + a];
    }
    if (Pchk) {
        std::vector<double> h(NW * cfn::N_CHK * cfn::SZ_PP);   // (packed triangle, cfnmpc_ws.hpp: pchk_at)
        HIP_TRY(hipMemcpy(h.data(), P.Pchk, h.size() * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < B; i++)
            for (int c = 0; c < cfn::N_CHK; c++) {
                if (cfn::chk_stage(c) >= (int)N) continue;   // no kernel fills a checkpoint at or behind the horizon's end: the caller's block stays untouched
                const double* pb = h.data() + ((i / 4) * cfn::N_CHK + c) * cfn::SZ_PP;
                for (int j = 0; j < 13; j++)
                    for (int r = 0; r < 13; r++)
                        Pchk[((i * cfn::N_CHK + c) * 13 + cfn::ext_of(r)) * 13 + cfn::ext_of(j)] = pb[cfn::pchk_at(j, (int)(i % 4), r)];
            }
    }
    if (status) HIP_TRY(hipMemcpy(status, P.status, B * sizeof(int), hipMemcpyDeviceToHost));
    return CFNMPC_OK;
}

#ifdef CFN_DEV
// development experiment (tools/sub_fleet_emul.py): the parts of one fused RTI step on separate streams.
// part 1: k_linfactor; part 2: everything behind it + the swap of the iterate buffers; flags bit 0: k_forward_half
int cfnmpc_debug_solve_part(cfnmpc_solver* s, int part, int flags, void* stream) {
    if (!s || s->P.fused != 1 || s->P.lbs) return CFNMPC_EINVAL;
    DeviceGuard dg(s);
    hipStream_t st = (hipStream_t)stream;
    if (part == 1) cfn::launch_linfactor(s->P, st);
    else {
        cfn::Params Q = s->P;
        Q.forward_half = flags & 1;
        cfn::launch_qp_start(Q, st, nullptr, true);
        cfn::launch_qp_ipm(Q, st);
        std::swap(s->P.xit, s->P.xitn);
        std::swap(s->P.uit, s->P.uitn);
        s->parity ^= 1;
    }
    HIP_TRY(hipGetLastError());
    return CFNMPC_OK;
}
#endif   // CFN_DEV

int cfnmpc_debug_linearise(cfnmpc_solver* s, void* stream) {
    if (!s) return CFNMPC_EINVAL;
    DeviceGuard dg(s);
    cfn::launch_linearise(s->P, s->chunks_all, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    s->lin_valid = true;
    return CFNMPC_OK;
}

int cfnmpc_debug_get_linearisation(cfnmpc_solver* s, double* A, double* Bm, double* b) {
    // Decodes the row-distributed stage blocks (AR, BR, b) into dense arrays in the EXTERNAL
    // state order: A [B][N][13][13], Bm [B][N][13][4], b [B][N][13].
    if (!s || !A || !Bm || !b) return CFNMPC_EINVAL;
    DeviceGuard dg(s);
    const cfn::Params& P = s->P;
    const size_t NW = ((size_t)P.NW / 16 + 1) * 16, N = P.N, B = P.B;   // (whole groups of 16 blocks)
    std::vector<double> ha(NW * N * cfn::SZ_A), hb(NW * N * cfn::SZ_B), hv(NW * N * cfn::SZ_V13);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(ha.data(), P.AR, ha.size() * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hb.data(), P.BR, hb.size() * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hv.data(), P.b, hv.size() * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < B; i++) {
        const size_t w = i / 4, q = i % 4;
        for (size_t k = 0; k < N; k++) {
            const size_t bs = ((w / 16) * N + k) * 16 + w % 16;   // [group of 16 blocks][stage][block of the group]
            const double* ab = ha.data() + bs * cfn::SZ_A;
            const double* bb = hb.data() + bs * cfn::SZ_B;
            const double* vb = hv.data() + bs * cfn::SZ_V13;
            double* Ad = A + (i * N + k) * 169;
            double* Bd = Bm + (i * N + k) * 52;
            for (int r = 0; r < 13; r++) {  // internal row / column indices
                for (int cc = 0; cc < 13; cc++) {
                    double val;
                    if (cc < 3) val = (r == cc) ? 1.0 : 0.0;
                    else {
                        const int sl = cc - 3;
                        val = r < cfn::ar_n(sl) ? ab[4 * cfn::ar_pre(sl) + q * cfn::ar_n(sl) + r] : 0.0;
                    }
                    Ad[cfn::ext_of(r) * 13 + cfn::ext_of(cc)] = val;
                }
                for (int a = 0; a < 4; a++) Bd[cfn::ext_of(r) * 4 + a] = bb[(a * 4 + q) * 13 + r];
                b[(i * N + k) * 13 + cfn::ext_of(r)] = vb[q * 13 + r];
            }
        }
    }
    return CFNMPC_OK;
}

int cfnmpc_debug_get_condensed(cfnmpc_solver* s, int block, double* H, double* D, int* m_out) {
    if (!s || !H || !D || !s->P.cond_N2 || block < 0 || block >= s->P.cond_N2) return CFNMPC_EINVAL;
    DeviceGuard dg(s);
    const cfn::Params& P = s->P;
    cfn::launch_linearise(P, s->chunks_all, nullptr);
    cfn::launch_pcond(P, nullptr);
    HIP_TRY(hipDeviceSynchronize());
    const int m = cfn::cond_len(P, block), mu = 4 * m, w = cfn::cond_w(m);
    const size_t slot = cfn::cb_size(cfn::cond_mmax(P)), B = P.B;
    std::vector<double> h(slot);
    // internal -> external order of the 13 state entries of z = (dU, dx, 1) and of the rows of D
    auto zext = [&](int i) { return (i >= mu && i < mu + 13) ? mu + cfn::ext_of(i - mu) : i; };
    for (size_t i = 0; i < B; i++) {
        HIP_TRY(hipMemcpy(h.data(), P.cb + (i * P.cond_N2 + block) * slot, slot * sizeof(double), hipMemcpyDeviceToHost));
        double* Hi = H + i * (size_t)w * w;
        double* Di = D + i * (size_t)13 * w;
        for (int r = 0; r < w; r++)
            for (int c = 0; c <= r; c++) {
                const double v = h[(size_t)r * (r + 1) / 2 + c];
                Hi[zext(r) * w + zext(c)] = v;
                Hi[zext(c) * w + zext(r)] = v;
            }
        const double* d = h.data() + cfn::cond_tri(w);
        for (int r = 0; r < 13; r++)
            for (int c = 0; c < w; c++) Di[cfn::ext_of(r) * w + zext(c)] = d[r * w + c];
    }
    if (m_out) *m_out = m;
    return CFNMPC_OK;
}

int cfnmpc_debug_get_viol(cfnmpc_solver* s, double* viol) {   // largest bound violation of the unconstrained minimiser (0: feasible)
    if (!s || !viol) return CFNMPC_EINVAL;
    DeviceGuard dg(s);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(viol, s->P.viol, (size_t)s->P.B * sizeof(double), hipMemcpyDeviceToHost));
    return CFNMPC_OK;
}

int cfnmpc_debug_get_head(cfnmpc_solver* s, int* head) {
    if (!s || !head) return CFNMPC_EINVAL;
    DeviceGuard dg(s);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(head, s->P.head, (size_t)s->P.B * sizeof(int), hipMemcpyDeviceToHost));
    return CFNMPC_OK;
}

// Work-list counts of the last step's constrained-QP phase (host array of four ints): constrained rows the compaction listed |
// rows listed for the interior-point fall-back (fleets that compact them: >= 16 S instances; else 0) | listed rows with heads of
// more than 16 stages | LATE rows of a split forward sweep (first violation behind stage 24, appended to the list by part two).
int cfnmpc_debug_get_list_counts(cfnmpc_solver* s, int* counts) {
    if (!s || !counts) return CFNMPC_EINVAL;
    DeviceGuard dg(s);
    HIP_TRY(hipDeviceSynchronize());
    int h[64];
    HIP_TRY(hipMemcpy(h, s->P.nipm, sizeof h, hipMemcpyDeviceToHost));
    counts[0] = h[0]; counts[1] = s->P.ipm_listed ? h[cfn::NI_LISTED] : 0; counts[2] = h[cfn::NI_LONG16]; counts[3] = s->P.fwd_split ? h[cfn::NI_LATE] : 0;
    return CFNMPC_OK;
}

}  // extern "C"
