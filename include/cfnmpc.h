/* cfnmpc.h -- batch C-ABI of the MI355X-native Crazyflie SQP-RTI solve engine (libcfnmpc.so).
 *
 * This is the batch superset of the reference's generated-solver boundary.
 * One `cfnmpc_solver` owns B independent NMPC instances (the reference owns exactly one, as
 * process globals: crazyflie_controller/src/acados_mpc.cpp:76-82).  The acados-named drop-in
 * (acados_create / acados_solve / ocp_nlp_*_set / ocp_nlp_out_get) lives in
 * include/acados_solver_crazyflie.h and is this engine with B = 1.
 *
 * Conventions
 *   - plain C, no torch / hip types in signatures; `stream` is a hipStream_t passed as void*
 *     (NULL = the default stream);
 *   - all arrays are FP64, AoS in the reference node's layouts:
 *       x  : [B][13]  = xq yq zq qw qx qy qz vbx vby vbz wx wy wz   (acados_mpc.cpp:117-131)
 *       u  : [B][4]   = w1..w4 [kRPM]                               (acados_mpc.cpp:133-138)
 *       yref   : [B][N][17], yref_e : [B][13]                        (acados_mpc.cpp:584-594)
 *   - `on_device`: 0 (CFNMPC_ON_HOST) host memory, copied synchronously; 2 (CFNMPC_ON_HOST_ASYNC) host memory,
 *     transfer only enqueued on `stream`; any other value: device memory of the solver's GPU (no copy through
 *     the host);
 *   - every call returns 0 on success, a negative CFNMPC_E* code otherwise; solver *status*
 *     per instance follows acados: 0 success, 2 max. iterations, 4 QP failure.
 *   - a solver (or fleet) lives on the HIP device that is current when it is created; later calls
 *     may be made with any device current (they select the solver's device for their duration);
 *   - there is NO CPU fallback: if no HIP device is usable, cfnmpc_create fails.
 */
#ifndef CFNMPC_H
#define CFNMPC_H

#ifdef __cplusplus
extern "C" {
#endif

#define CFNMPC_NX 13
#define CFNMPC_NU 4
#define CFNMPC_NY 17
#define CFNMPC_NYN 13

#define CFNMPC_OK 0
#define CFNMPC_EINVAL (-1)  /* bad argument                              */
#define CFNMPC_EHIP (-2)    /* HIP runtime error / no device             */
#define CFNMPC_ENOMEM (-3)  /* device allocation failed                  */

/* values of the `on_device` argument of the array setters / getters */
#define CFNMPC_ON_HOST 0        /* host pointer; the call returns when the transfer is complete                       */
#define CFNMPC_ON_DEVICE 1      /* device pointer of the solver's GPU (any non-zero value other than 2 historically)   */
#define CFNMPC_ON_HOST_ASYNC 2  /* host pointer; the transfer is only ENQUEUED on `stream`: the caller keeps the array
                                   alive and synchronises the stream before it reads (getters) or reuses (setters) it --
                                   what cfnmpc_multi_* uses to overlap the I/O of its shards                            */

#define CFNMPC_INIT_ACADOS 0 /* x_k = [0,0,0,1,0..], u_k = 0 (generate_c_code.py:135) */
#define CFNMPC_INIT_HOVER 1  /* x_k = current x0, u_k = hover speed                                      */

typedef struct cfnmpc_solver cfnmpc_solver;

/* ABI guard.  cfnmpc_opts has grown with every round; a caller built against an older header (or a hand-written binding
 * that stops short of the last field) must be refused instead of being written past:
 *   - CFNMPC_ABI_VERSION changes whenever cfnmpc_opts or a signature below changes; cfnmpc_abi_version() returns the
 *     LIBRARY's value, cfnmpc_opts_size() its sizeof(cfnmpc_opts);
 *   - cfnmpc_default_opts_v(opts, sizeof *opts) fills `opts` only if the caller's size is the library's (CFNMPC_EINVAL
 *     otherwise, nothing written) -- what C callers and bindings should use; CFNMPC_DEFAULT_OPTS(&o) spells it;
 *   - every cfnmpc_opts starts with its own size (set by cfnmpc_default_opts*), and cfnmpc_create / cfnmpc_fleet_create /
 *     cfnmpc_multi_create* refuse (CFNMPC_EINVAL) an object whose struct_size is not the library's. */
#define CFNMPC_ABI_VERSION 9

/* Replaces the constants baked into the generated solver by
 * crazyflie_controller/scripts/crazyflie_full_model/generate_c_code.py:41-146. */
typedef struct cfnmpc_opts {
    int struct_size;     /* sizeof(cfnmpc_opts) as the filler saw it (ABI guard above); do not change             */
    int N;               /* horizon length, generate_c_code.py:42 (50); 5 <= N <= 4096 (FP64 Riccati: with an
                            iterate far from the reference over the whole horizon the costate grows with N and
                            the stationarity residual with it -- 1e-12 at N = 50, 2e-10 at 200, 3.5e-7 at 4096) */
    double dt;           /* shooting interval Tf/N, generate_c_code.py:41-42 (0.015)       */
    double W[CFNMPC_NY]; /* diag of stage weight W, generate_c_code.py:63-84               */
    double WN[CFNMPC_NYN]; /* diag of terminal weight W_e = 50 Q, generate_c_code.py:109   */
    double u_min, u_max; /* input box, generate_c_code.py:133-134 (0, 22)                  */
    double tol;          /* QP: max-norm tolerance on all residuals (1e-8)                 */
    int max_iter;        /* QP: interior-point iteration cap (50)                          */
    double tau;          /* QP: fraction to the boundary (0.995)                           */
    double thr0;         /* QP: slack floor of the starting point (1.0)                    */
    double lam0_min;     /* QP: complementarity floor of the starting point (1e-2)         */
    double mu0_scale;    /* QP: starting complementarity = max(lam0_min, mu0_scale * largest
                            bound violation of the unconstrained minimiser) (0.1)          */
    int active_horizon;  /* QP: 1 = interior-point sweeps only over the head of the horizon
                            whose inputs can saturate; the unconstrained tail keeps its
                            Riccati feedback law and is verified afterwards (exact); 0 = all
                            N stages in every sweep (default 1)                              */
    double ah_margin;    /* active horizon: an unconstrained input closer to a bound than this
                            fraction of (u_max - u_min) counts as 'tight' (0.10)              */
    int ah_extra;        /* active horizon: stages added after the last tight stage (4)       */
    int active_set;      /* QP: 1 (default) = solve the box-constrained QP by a primal-dual active-set
                            iteration first: guess the active input bounds from the unconstrained
                            minimiser, solve the equality-constrained QP (one Riccati factorisation
                            with the active inputs fixed), check signs of the multipliers and bounds
                            of the free inputs, repeat until the set is stationary -- which proves
                            the KKT conditions of the strictly convex QP, i.e. the exact solution
                            (typically 1-3 solves instead of 5-10 interior-point iterations).  Falls
                            back to the interior-point iteration if the set does not settle within
                            12 solves.  0 = interior point only (the reference's QP method class).
                            cfnmpc_get_stats reports active-set solves + interior-point iterations. */
    int forward_sweep;   /* start solve, forward sweep: 1 = matrix-free, one instance per lane (dx+ = A dx + B du + b
                            as the directional derivative of the RK4 map: fewest bytes, B / 64 wavefronts);
                            2 = on the stored (A, B, b), four instances per wavefront (16 x more wavefronts, a
                            shorter dependent chain per stage: faster while the fleet is too small to fill the
                            GPU); 0 (default) = by fleet size in waves per SIMD (2 below 6 S instances, S = the device's
                            SIMD count, 1024 on MI355X: measured cross-over against the split matrix-free sweep, profiles/r05_forward_split.md).  Same results to
                            rounding.                                                                        */
    int cond_N2;         /* QP: partial condensing (PARTIAL_CONDENSING_HPIPM, generate_c_code.py:140; acados'
                            qp_solver_cond_N, which the generator leaves at its default): 0 or N (default 0) =
                            none -- the Riccati sweeps run over the N original stages; 0 < cond_N2 < N =
                            the N stages are regrouped into cond_N2 blocks of consecutive stages (the first
                            N mod cond_N2 one stage longer; at most 10 stages per block), the interior states
                            of each block are eliminated exactly (`pcond`: a QP with cond_N2 stages, 13
                            states, 4 x block-length inputs), that QP is solved by the Riccati recursion with
                            one dense Cholesky factorisation per block -- interior point on the same
                            condensed blocks for the instances whose unconstrained minimiser leaves the box
                            -- and the solution is expanded through the original stage dynamics.  Same primal
                            solution (strictly convex QP); active_set / active_horizon do not apply.
                            Measured slower than the uncondensed path at every N2:
                            an option for parity with the reference's solver plan, not the default.        */
    int step_graph;      /* 1: cfnmpc_solve replays the launches of an RTI step from a captured hipGraph (one per
                            parity of the two iterate buffers; re-captured after cfnmpc_set_weights / _set_box)
                            instead of launching its 7-8 kernels one by one;
                            steps timed with cfnmpc_set_profiling are launched individually.  0 (default): the
                            kernels already run back to back (18 us of gaps per step at 4096 instances), the
                            graph saves about half of that.                          */
    int as_passes;       /* QP, active_set = 1: how the active-set solves of the constrained instances are scheduled.
                            -1: one monolithic kernel on a wave-blocked compact copy (four instances per wavefront stay
                            together until the slowest has settled, been rolled out and verified in-wave);
                            -3: the same kernel's solves only, followed by a COMMIT kernel: new iterate = the start
                            solve's candidate + delta -- head stages element-wise from the solve's own du / dx, the tail
                            through the closed loop of the unconstrained feedback law, whose inputs are verified against
                            the box there (rows whose tail leaves it are solved again over a longer head);
                            0 (default): -3 below 36 S instances (20 S for N <= 40; S = the device's SIMD count, 1024 on MI355X),
                            -1 beyond -- measured on MI355X, profiles/r04_thresholds.md, DESIGN.md section 5.5.  Same solves in
                            either mode; results agree to rounding.  Any other value: CFNMPC_EINVAL (round 3's level-synchronous
                            passes and instance-contiguous store, -2 / 1..12, were slower at every fleet size and live in the
                            development build only since ABI 9). */
    double ipm_clip_viol; /* QP, interior point: CLIPPED START when the unconstrained minimiser leaves the box by more than
                            this many box widths (2.0; 0 = never).  From such a point (vehicles far from their iterate's
                            trajectory: ~100 kRPM outside and more) the infeasible start spends 30 - 60 iterations at tiny
                            step lengths; instead the iteration starts inside the box -- inputs clipped to a margin of
                            ipm_clip_margin of the width -- with multipliers that absorb the gradient of the condensed QP
                            there (one forward + one backward costate sweep) and a complementarity floor scaled by it.
                            Captured fall-back QPs: 63 -> 22 iterations, none at the cap any more; ordinary QPs are
                            better served by the infeasible start (5 against 8 iterations), hence the threshold.          */
    double ipm_clip_margin; /* ... 0.05                                                                                   */
    double as_skip_viol; /* QP, active_set = 1: instances whose unconstrained minimiser leaves the box by more than this many
                            box widths skip the active-set iteration and go straight to the interior point (4.0; 0 = never).
                            Measured on the engine's closed loops at 2x / 3x the bench's disturbances: the iteration settles
                            for 68 - 71 % of the instances between 2 and 4 widths, 14 - 15 % between 4 and 8, 1 - 2 % beyond --
                            twelve futile solves per instance otherwise.                                                 */
    int reinit_failed;   /* 1: an instance whose previous RTI step ended in status 4 (QP failure: the factorisation around its
                            iterate is not positive definite / not finite) restarts the next step from x_k = its current x0,
                            u_k = the input reference of its stage, instead of re-linearising around the same iterate and failing
                            the same way for good.  0 (default): the reference's behaviour -- the node ignores the status and keeps
                            the iterate (acados_mpc.cpp:611-616).                                                        */
    int start_solve;     /* start solve (linearisation + backward Riccati factorisation of the unconstrained QP):
                            1 = two kernels with the stage blocks (A, B, b) stored in between (k_linearise writes 162 doubles
                            per instance and stage, k_factor reads them back -- half of the bytes an RTI step moves);
                            2 = FUSED (k_linfactor): the factorisation's wavefront linearises each stage itself on its way
                            backward (every lane integrates one sensitivity column, the columns reach the row form through
                            the LDS tile of the W transpose) and (A, B, b) never touch HBM; the constrained instances alone
                            are re-linearised straight into the compact store of their QP kernels (k_linearise_clist);
                            3 = validation: the fused kernel for the factorisation, the stored blocks for everything behind it;
                            0 (default) = 1: measured on MI355X the fused kernel takes the step's HBM traffic from 17.8 to 8.0 GB
                            and is bound by its vector instructions instead (k_linfactor 2.85 - 2.99 ms against 2.70 - 2.87 ms for
                            the pair).  2 is refused (CFNMPC_EINVAL) together with what reads the stored
                            blocks: cond_N2, forward_sweep = 2, as_passes other than 0 / -1; it switches the
                            automatic choices to the matrix-free forward sweep and the monolithic active-set kernel, and a solver
                            that is given per-stage boxes later runs the stored-block kernels while they are set.  Same results
                            to rounding (tests/test_gpu_linfactor.py). */
    int as_warm;         /* QP, active_set = 1: 1 = WARM START of the active set.  An instance whose previous RTI step ended in a
                            settled active-set solve starts its first solve from the union of that solve's final set and the
                            violations of today's unconstrained minimiser, instead of the violations alone (consecutive steps of
                            a constrained vehicle share most of their active set, and the reference never shifts its iterate --
                            acados_mpc.cpp:581-611 -- so the classes are taken stage for stage).  Exactness is untouched: whatever
                            the start, the iteration ends in a stationary classification, i.e. the KKT point of the strictly
                            convex QP; only the number of solves changes.  The state (4 N bytes + a flag per instance) is cleared by
                            cfnmpc_init_iterate / cfnmpc_set_iterate.  Read by the monolithic and the solves + commit kernels
                            (as_passes 0 / -1 / -3); the level-synchronous variants ignore it.  Default 0: see DESIGN.md section 5.5
                            for the measured solve histograms. */
    int as_dense;        /* QP, active_set = 1, solves + commit structure (as_passes 0 auto / -3): rows whose head is at most 16
                            stages long (99 % of the constrained rows of the bench workload) are solved on the HEAD-CONDENSED DENSE
                            QP -- the same active-set iteration on H = R + Gamma' Q Gamma of the head (4 x head inputs, the tail's
                            cost-to-go as terminal weight), inverted once per row, every solve then a system of the size of the
                            active set (csrc/cfnmpc_asdense.hip) -- instead of one Riccati factorisation + forward sweep over the
                            head per solve.  Same classification rule, same solve counts, same solutions to rounding (the reference's
                            own plan condenses too: PARTIAL_CONDENSING_HPIPM, generate_c_code.py:140).  1 = on (selects the solves +
                            commit structure), -1 = off, 0 (default) = by measurement.  Scalar box only
                            (while per-stage boxes are set the monolithic kernels run).  An explicit 1 that cannot be honoured --
                            together with as_warm, active_set = 0, cond_N2, start_solve = 2 or as_passes other than 0 / -3 -- is
                            refused (CFNMPC_EINVAL), not dropped. */
    int forward_split;   /* start solve, matrix-free forward sweep inside the as_dense structure: 1 = in TWO launches -- stages [0, 24)
                            with the classification of the instances over that window, the compaction, and stages [24, N) of every
                            instance BESIDE the constrained rows' kernels (their heads of at most 24 stages read nothing behind
                            stage 24; rows with longer heads follow the second part on its stream) -- half of the sweep's 50-stage
                            latency chain leaves the critical path of a small fleet's step.  An instance that is feasible over the
                            first window and leaves the box behind it joins the list late and is solved with the retry kernel.
                            Same SOLUTIONS (candidates bitwise; a row's head class and its violation measure -- what as_skip_viol
                            and ipm_clip_viol compare -- come from the first window only, so a row may take another route than
                            with the single launch: a retry over a longer head after the tail verification, the active-set
                            iteration instead of the interior point or the other way round; either route ends at the QP's one
                            solution, to its own accuracy).  -1 = off, 0 (default) = by measurement.  An explicit 1 that cannot
                            be honoured -- outside the as_dense structure, N < 40, forward_sweep = 2, cond_N2, start_solve 2 / 3 -- is
                            refused (CFNMPC_EINVAL). */
} cfnmpc_opts;

void cfnmpc_default_opts(cfnmpc_opts *opts);   /* unchecked: the caller's struct MUST be this header's */
int cfnmpc_default_opts_v(cfnmpc_opts *opts, int sizeof_opts);
int cfnmpc_opts_size(void);
int cfnmpc_abi_version(void);
#define CFNMPC_DEFAULT_OPTS(o) cfnmpc_default_opts_v((o), (int)sizeof(cfnmpc_opts))

/* acados_create() / acados_free() equivalents (acados_mpc.cpp:225,418) for B instances on the
 * calling thread's current HIP device. */
int cfnmpc_create(cfnmpc_solver **out, int batch, const cfnmpc_opts *opts);
int cfnmpc_free(cfnmpc_solver *s);
int cfnmpc_batch(const cfnmpc_solver *s);
int cfnmpc_horizon(const cfnmpc_solver *s);
/* bytes of device workspace held by the solver */
unsigned long long cfnmpc_workspace_bytes(const cfnmpc_solver *s);

/* ocp_nlp_constraints_model_set(.., 0, "lbx"/"ubx", x0) equivalent (acados_mpc.cpp:581-582) */
int cfnmpc_set_x0(cfnmpc_solver *s, const double *x0, int on_device, void *stream);
/* ocp_nlp_cost_model_set(.., k, "yref", ..) for k = 0..N (acados_mpc.cpp:590-594) */
int cfnmpc_set_yref(cfnmpc_solver *s, const double *yref, const double *yref_e, int on_device, void *stream);
/* Reference windows generated ON THE DEVICE with the reference node's state machine
 * (NMPC::iteration, acados_mpc.cpp:430-516) -- replaces the host-side fill of yref_sign and the
 * N+1 ocp_nlp_cost_model_set("yref") calls per step (:590-594) for fleets:
 *   mode[i] = 0 Regulation (rows = des_xyz[i], identity attitude, zeros, uss), 1 Tracking (rows
 *   iter[i]..iter[i]+N of traj, then ++iter[i]; becomes 2 once iter[i] >= n_rows - N, keeping the
 *   previous window for that step), 2 Position_Hold (xyz of the last trajectory row).
 * All pointers are DEVICE pointers: traj [n_rows][17] (may be NULL with n_rows = 0 if no instance
 * tracks), mode [B] / iter [B] (int, updated in place), des_xyz [B][3]. */
int cfnmpc_set_yref_windows(cfnmpc_solver *s, const double *traj, int n_rows, int *mode, int *iter,
                            const double *des_xyz, double uss, void *stream);
/* ocp_nlp_cost_model_set(.., k, "W", ..) equivalent (acados_mpc.cpp:596-602, compiled out by
 * SET_WEIGHTS 0 in the reference): diagonal stage / terminal weights for ALL instances and
 * stages; either pointer may be NULL (unchanged).  State weights must be >= 0 (the reference's
 * dynamic_reconfigure ranges start at 0.0, config/crazyflie_params.cfg), input weights > 0; the
 * arrays are validated as a whole before anything is changed. */
int cfnmpc_set_weights(cfnmpc_solver *s, const double *W /*[17]*/, const double *WN /*[13]*/);
/* ocp_nlp_constraints_model_set(.., k, "lbu"/"ubu", ..) equivalent (acados_mpc.cpp:605-608, compiled
 * out by FIXED_U0 0 at :111) for the case the engine supports: ONE input box [u_min, u_max] for all
 * inputs, stages and instances (generate_c_code.py:133-134).  Takes effect at the next
 * cfnmpc_solve.  Per-stage, per-input boxes: cfnmpc_set_box_stages below. */
int cfnmpc_set_box(cfnmpc_solver *s, double u_min, double u_max);
/* Per-stage, per-input box: lb, ub [B][N][4] (what "lbu" / "ubu" on INDIVIDUAL stages set in acados -- the reference's
 * FIXED_U0 variant pins stage 0 to the input in flight, lbu = ubu = u1, acados_mpc.cpp:605-608).  lb[i] = ub[i] makes
 * that input an equality (the active-set solves keep it fixed whatever its multiplier's sign; the interior-point
 * fall-back needs lb < ub).  HOST arrays are validated (lb <= ub everywhere, no NaN: CFNMPC_EINVAL, nothing changed); DEVICE
 * arrays are taken as they are -- the caller guarantees lb <= ub (an inverted box ends in status 4 for that vehicle).  The
 * call is atomic: on any failure the previous boxes (both arrays) stay in force.  NULL, NULL returns to the scalar box of
 * cfnmpc_set_box.  Steps with per-stage boxes run
 * the row-group forward sweep and the monolithic QP kernels (instantiated for them); not with cond_N2. */
int cfnmpc_set_box_stages(cfnmpc_solver *s, const double *lb, const double *ub, int on_device, void *stream);

/* Iterate (the persistent nlp_out of acados_mpc.cpp:77): initial guess, save / restore. */
int cfnmpc_init_iterate(cfnmpc_solver *s, int mode, void *stream);
int cfnmpc_set_iterate(cfnmpc_solver *s, const double *x /*[B][N+1][13]*/, const double *u /*[B][N][4]*/, int on_device, void *stream);
int cfnmpc_get_iterate(cfnmpc_solver *s, double *x, double *u, int on_device, void *stream);

/* acados_solve() equivalent (acados_mpc.cpp:611): n_rti consecutive SQP-RTI steps
 * (linearise -> Riccati interior-point QP -> full step) for all B instances, asynchronous on
 * `stream`.  The reference calls it with one step. */
int cfnmpc_solve(cfnmpc_solver *s, int n_rti, void *stream);

/* One complete control step from HOST buffers with a single synchronisation -- what the node
 * does per sample (acados_mpc.cpp:581-625: set lbx/ubx, N+1 yref rows, acados_solve(), read u/x):
 * x0 [B][13], yref [B][N][17], yref_e [B][13] in; one RTI step; the whole iterate out
 * (u [B][N][4], x [B][N+1][13]) with status, QP solve count and residual per instance.  The
 * transfers go through pinned staging memory owned by the solver, asynchronously on `stream`,
 * which is synchronised once before returning.  Any output pointer may be NULL. */
int cfnmpc_step_host(cfnmpc_solver *s, const double *x0, const double *yref, const double *yref_e,
                     double *u, double *x, int *status, int *qp_iter, double *res, void *stream);

/* ocp_nlp_out_get(.., stage, "u"/"x", ..) equivalents (acados_mpc.cpp:619-625) */
int cfnmpc_get_u(cfnmpc_solver *s, int stage, double *u /*[B][4]*/, int on_device, void *stream);
int cfnmpc_get_x(cfnmpc_solver *s, int stage, double *x /*[B][13]*/, int on_device, void *stream);
/* status (acados_solve() return value), QP iteration count, nlp_out->inf_norm_res stand-in
 * (max-norm residual of the last QP).  Any pointer may be NULL.
 * qp_iter: 0 = the unconstrained minimiser was feasible; otherwise the number of active-set solves of a row they settled (a row
 * of the large fleets' kernel that was solved AGAIN over a longer head -- its own or a wave-mate's tail had left the box -- counts
 * the solves of all attempts; the later ones start from the previous attempt's final set and take one or two), or
 * the interior-point ITERATIONS of a row that fell back (after 12 unsettled solves, or skipped by as_skip_viol; such a row runs
 * the interior point over all N stages) -- the two ranges overlap (twelve iterations are not twelve solves).  res tells them apart: exactly 0.0 for a row settled by active-set
 * solves (a stationary classification IS the KKT system), the interior point's final residual (> 0, <= tol at status 0)
 * for a row it solved. */
int cfnmpc_get_stats(cfnmpc_solver *s, int *status /*[B]*/, int *qp_iter /*[B]*/, double *res /*[B]*/, int on_device, void *stream);

/* Output stage of the reference node for the whole fleet, on the device (NMPC::iteration,
 * acados_mpc.cpp:619-670): from the current iterate (u0 = inputs of stage 0, u1 = stage 1, x4 =
 * state of stage 4) it forms what the node publishes per vehicle,
 *   motvel  [B][4] int32 : u0 truncated toward zero -- /crazyflie/acados_motvel, the int32 fields of
 *                          PropellerSpeedsStamped (msg/PropellerSpeedsStamped.msg:2-5, :637-640);
 *   cmd_vel [B][4] double: linear.x = pitch [deg] = +deg(theta), linear.y = roll [deg] = -deg(phi),
 *                          linear.z = thrust PWM = (int)((mean(u1)*1000 - 4070.3)/0.2685) (:421-425),
 *                          angular.z = yaw rate [deg/s] = deg(x4.wz); (phi, theta) from the normalised
 *                          quaternion of x4 with the node's formulas (:384-404, :645-668).
 * motvel may be NULL.  on_device as everywhere. */
int cfnmpc_get_cmd(cfnmpc_solver *s, double *cmd_vel /*[B][4]*/, int *motvel /*[B][4]*/, int on_device, void *stream);

/* Per-phase timing (the role of nlp_out->total_time, acados_mpc.cpp:616): when enabled,
 * cfnmpc_solve brackets its two phases (linearisation, QP) with HIP events on the launch stream;
 * cfnmpc_get_profile waits for them and returns the average durations [ms] over the RTI steps
 * since the last call (at most 4096 steps are timed between two calls), then resets.  */
int cfnmpc_set_profiling(cfnmpc_solver *s, int enable);
int cfnmpc_get_profile(cfnmpc_solver *s, double *ms_linearise, double *ms_qp, int *n_steps);
/* The same timed steps split per kernel group (seven events per step): ms[6] = linearisation | start solve backward
 * (k_factor) | start solve forward (k_forward / k_forward_rg + k_rank) | compaction (k_compact + k_scatter) |
 * active-set kernels (k_as, or the passes + commit + retry) | interior point for what they left (k_ipm_rest / k_ipm).
 * Partial-condensing steps report their phases in ms[0] / ms[5] only.  Resets like cfnmpc_get_profile
 * (call one of the two). */
int cfnmpc_get_profile_kernels(cfnmpc_solver *s, double *ms, int *n_steps);
/* ... and per timed step instead of averaged (the active-set kernels of a step in which a tail check fails take four times the
 * average: a mean hides what a deadline sees): ms_steps [max_steps][6], *n_steps = steps written; resets likewise */
int cfnmpc_get_profile_steps(cfnmpc_solver *s, double *ms_steps, int max_steps, int *n_steps);

/* crazyflie_acados_sim_solve() equivalent, batched (acados_estimator.cpp:573-593):
 * xn = RK4(x, u) over T seconds in `steps` sub-steps.  Stateless. */
int cfnmpc_sim(int batch, const double *x, const double *u, double T, int steps, double *xn, int on_device, void *stream);

/* ESTIMATOR::predictor() for a fleet (acados_estimator.cpp:521-634), DEVICE pointers only:
 * assembles the 13-state from mocap position, onboard Euler angles [deg, as published by the
 * driver; the pitch sign flip of :495 is applied inside] and gyro rates, estimates the world
 * velocity with the reference's low-pass filter (:356-368; the reference always takes this branch
 * because it passes the absolute start time as "elapsed time"; use_lpf = 0 selects the finite
 * difference), rotates it into the body frame (:414-440) and integrates the model over `delay`
 * with the latest inputs (:573-593).  meas [B][9] = x y z roll pitch yaw wx wy wz; filt [B][9] =
 * previous position, v[k-1], v[k-2] (updated in place); u [B][4]; outputs x_est, x_pred [B][13]. */
int cfnmpc_estimate(int batch, const double *meas, double *filt, const double *u, double dt, int use_lpf,
                    double delay, int steps, double *x_est, double *x_pred, void *stream);

/* Kernel-level access for parity tests (oracle comparison of the linearisation): copies the
 * stage blocks of the last linearisation as dense row-major arrays in the reference's state
 * order: A [B][N][13][13], Bm [B][N][13][4], b [B][N][13] (host pointers). */
int cfnmpc_debug_get_linearisation(cfnmpc_solver *s, double *A, double *Bm, double *b);
/* runs only the linearisation kernel */
int cfnmpc_debug_linearise(cfnmpc_solver *s, void *stream);
/* start solve, backward half only (parity tests of the fused kernel, timing): mode 1 = k_linearise + k_factor, 2 = k_linfactor;
 * `reps` repetitions, *ms (may be NULL) = average duration of one, HIP events on `stream` */
int cfnmpc_debug_start_factor(cfnmpc_solver *s, int mode, int reps, double *ms, void *stream);
/* what that sweep leaves behind, as dense host arrays in the reference's state order (any pointer may be NULL):
 * K [B][N][4][13], d [B][N][4], cost-to-go checkpoints Pchk [B][6][13][13] (stages 4, 8, 12, 16, 24, 32 below N), status [B] */
int cfnmpc_debug_get_factor(cfnmpc_solver *s, double *K, double *d, double *Pchk, int *status);
/* partial condensing (cond_N2 > 0): runs linearisation + pcond and copies condensed block `block`
 * of every instance to the host as dense arrays in the reference's state order, with
 * z = (dU (4 m), dx (13), 1), w = 4 m + 14, m = stages of that block:
 *   H [B][w][w] symmetric (cost 1/2 z'H z; the [w-1][w-1] entry is not defined), D [B][13][w]
 *   (dx at the start of the next block = D z).  *m_out receives m. */
int cfnmpc_debug_get_condensed(cfnmpc_solver *s, int block, double *H, double *D, int *m_out);
/* number of leading stages the last QP's interior-point sweeps covered, per instance [B] (host) */
int cfnmpc_debug_get_viol(cfnmpc_solver *s, double *viol /*[B]*/);
int cfnmpc_debug_get_head(cfnmpc_solver *s, int *head);
/* work-list counts of the last step (host, four ints): constrained rows listed | rows listed for the interior-point fall-back
 * (fleets that compact them; else 0) | listed rows with heads of more than 16 stages | late rows of a split forward sweep */
int cfnmpc_debug_get_list_counts(cfnmpc_solver *s, int *counts /*[4]*/);

/* ---- mixed-horizon fleets (BASELINE.json config C5: N in {30, 50, 100} side by side) ----------
 * The reference fixes N when the solver is generated (generate_c_code.py:41-42; one process per
 * vehicle, acados_mpc.cpp:76-82).  A fleet takes one horizon PER VEHICLE, buckets the vehicles by
 * horizon (one cfnmpc_solver per distinct N) and keeps the caller's vehicle order at the
 * boundary: rows move between that order and the buckets on the device (device pointers) or on
 * the host (host pointers).  Buckets are solved concurrently, each on an internal stream forked
 * from and joined to `stream`.  opts->N is ignored.  Layouts:
 *     x0 [B][13];  yref [B][Nmax][17] (vehicle i uses rows 0..N_i-1), yref_e [B][13];
 *     get_u: stage < min N;  get_x: stage <= min N;  stats [B]. */
typedef struct cfnmpc_fleet cfnmpc_fleet;
int cfnmpc_fleet_create(cfnmpc_fleet **out, int batch, const int *N_per_instance /*[B]*/, const cfnmpc_opts *opts);
int cfnmpc_fleet_free(cfnmpc_fleet *f);
int cfnmpc_fleet_batch(const cfnmpc_fleet *f);
int cfnmpc_fleet_min_horizon(const cfnmpc_fleet *f);
int cfnmpc_fleet_max_horizon(const cfnmpc_fleet *f);
int cfnmpc_fleet_num_buckets(const cfnmpc_fleet *f);
/* bucket b (ascending N): its horizon, size, solver (owned by the fleet; for per-bucket calls such
 * as cfnmpc_get_iterate) and the fleet index of each of its rows [count].  Any pointer may be NULL. */
int cfnmpc_fleet_bucket(const cfnmpc_fleet *f, int bucket, int *N, int *count, cfnmpc_solver **solver, int *index);
unsigned long long cfnmpc_fleet_workspace_bytes(const cfnmpc_fleet *f);
int cfnmpc_fleet_set_x0(cfnmpc_fleet *f, const double *x0, int on_device, void *stream);
int cfnmpc_fleet_set_yref(cfnmpc_fleet *f, const double *yref, const double *yref_e, int on_device, void *stream);
int cfnmpc_fleet_set_weights(cfnmpc_fleet *f, const double *W /*[17]*/, const double *WN /*[13]*/);
int cfnmpc_fleet_set_box(cfnmpc_fleet *f, double u_min, double u_max);
/* cfnmpc_set_box_stages for a fleet: HOST arrays [B][Nmax][4] in the fleet's vehicle order (rows behind a vehicle's own
 * horizon are ignored); NULL, NULL: back to the scalar box */
int cfnmpc_fleet_set_box_stages(cfnmpc_fleet *f, const double *lb, const double *ub);
int cfnmpc_fleet_init_iterate(cfnmpc_fleet *f, int mode, void *stream);
int cfnmpc_fleet_solve(cfnmpc_fleet *f, int n_rti, void *stream);
int cfnmpc_fleet_get_u(cfnmpc_fleet *f, int stage, double *u /*[B][4]*/, int on_device, void *stream);
int cfnmpc_fleet_get_x(cfnmpc_fleet *f, int stage, double *x /*[B][13]*/, int on_device, void *stream);
int cfnmpc_fleet_get_cmd(cfnmpc_fleet *f, double *cmd_vel /*[B][4]*/, int *motvel /*[B][4]*/, int on_device, void *stream);
int cfnmpc_fleet_get_stats(cfnmpc_fleet *f, int *status, int *qp_iter, double *res, int on_device, void *stream);

/* ---- one fleet across several GPUs of a node, from ONE process --------------------------------
 * The reference owns one vehicle per process (acados_mpc.cpp:76-82); instances are independent, so a
 * fleet splits into contiguous shards with no exchange between devices: shard i
 * = vehicles [lo_i, hi_i) on HIP device device_ids[i] (ids may repeat), each with its own solver and
 * stream.  cfnmpc_multi_solve launches every shard's step and returns without waiting; getters wait
 * for what they read.  Host arrays cover the WHOLE fleet ([B][..] as in the single-device calls).
 * Device-resident I/O: take a shard's solver / stream with cfnmpc_multi_shard and use the
 * single-device calls with pointers of that device. */
typedef struct cfnmpc_multi cfnmpc_multi;
int cfnmpc_multi_create(cfnmpc_multi **out, int n_shards, const int *device_ids /*[n_shards]*/, int total_batch,
                        const cfnmpc_opts *opts);
/* Mixed horizons across GPUs (BASELINE.json config C5 "8 x MI355X, divergent-length stress"; SURVEY.md section 8e: "first
 * bucket by N, then balance buckets across GPUs by sum N_i"): one horizon PER VEHICLE; the vehicles are dealt out over the
 * shards by cfnmpc_shard_by_horizon, every shard is a cfnmpc_fleet (one solver per horizon bucket) over its -- NOT
 * contiguous -- index set.  Host arrays cover the whole fleet in the caller's order with the fleet layouts (yref
 * [B][Nmax][17], boxes [B][Nmax][4], get_u: stage < min N).  cfnmpc_multi_shard does not apply (CFNMPC_EINVAL); use
 * cfnmpc_multi_shard_fleet: the shard's fleet, its vehicle count and their indices [count] in the caller's order. */
int cfnmpc_multi_create_horizons(cfnmpc_multi **out, int n_shards, const int *device_ids /*[n_shards]*/, int total_batch,
                                 const int *N_per_instance /*[total_batch]*/, const cfnmpc_opts *opts);
int cfnmpc_multi_shard_fleet(const cfnmpc_multi *m, int shard, cfnmpc_fleet **fleet, int *count, int *index, int *device, void **stream);
/* The partitioner itself (pure host code, no device needed; also what bench.py's ranks use for `--workload mixed`):
 * vehicles in order of decreasing horizon, each to the shard with the smallest sum N so far (ties: lowest shard).
 * shard_of [batch] receives the shard of every vehicle; sum N of any two shards differs by at most one vehicle's horizon. */
int cfnmpc_shard_by_horizon(int batch, const int *N_per_instance, int n_shards, int *shard_of);
int cfnmpc_multi_free(cfnmpc_multi *m);
int cfnmpc_multi_batch(const cfnmpc_multi *m);
int cfnmpc_multi_num_shards(const cfnmpc_multi *m);
int cfnmpc_multi_shard(const cfnmpc_multi *m, int shard, cfnmpc_solver **solver, int *lo, int *hi, int *device, void **stream);
int cfnmpc_multi_set_x0(cfnmpc_multi *m, const double *x0 /*[B][13] host*/);
int cfnmpc_multi_set_yref(cfnmpc_multi *m, const double *yref /*[B][N][17] host*/, const double *yref_e /*[B][13] host*/);
int cfnmpc_multi_set_weights(cfnmpc_multi *m, const double *W, const double *WN);
int cfnmpc_multi_init_iterate(cfnmpc_multi *m, int mode);
int cfnmpc_multi_solve(cfnmpc_multi *m, int n_rti);   /* asynchronous on every shard's stream */
int cfnmpc_multi_sync(cfnmpc_multi *m);
int cfnmpc_multi_get_u(cfnmpc_multi *m, int stage, double *u /*[B][4] host*/);
int cfnmpc_multi_get_x(cfnmpc_multi *m, int stage, double *x /*[B][13] host*/);
int cfnmpc_multi_get_cmd(cfnmpc_multi *m, double *cmd_vel /*[B][4] host*/, int *motvel /*[B][4] host or NULL*/);
int cfnmpc_multi_get_stats(cfnmpc_multi *m, int *status, int *qp_iter, double *res);
/* cfnmpc_set_box / cfnmpc_set_box_stages (host arrays [B][N][4] of the whole fleet) for every shard */
int cfnmpc_multi_set_box(cfnmpc_multi *m, double u_min, double u_max);
int cfnmpc_multi_set_box_stages(cfnmpc_multi *m, const double *lb, const double *ub);

const char *cfnmpc_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CFNMPC_H */
