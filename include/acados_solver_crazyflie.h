/* acados_solver_crazyflie.h -- drop-in replacement of the GENERATED acados solver header that
 * crazyflie_controller/src/acados_mpc.cpp:72-73 includes ("acados_solver_crazyflie.h",
 * "crazyflie_model/crazyflie_model.h") plus the handful of acados_c entry points the node calls
 * (acados_mpc.cpp:61-69).  Implemented by libacados_solver_crazyflie.so, which is the batch
 * engine of include/cfnmpc.h with batch = 1 on the calling thread's HIP device.
 *
 * Source-level compatibility, old capsule-less template: the real headers are
 * not under /root/reference (empty acados submodule, git-ignored c_generated_code/), so the
 * structs below are this library's own; only the members the node reads directly are promised
 * (nlp_out->inf_norm_res, nlp_out->total_time: acados_mpc.cpp:615-616).
 *
 * Threading / errors as in the reference: single-threaded, process-global solver, int status
 * only (0 success, 1 failure, 2 max. iterations, 4 QP failure), never aborts the process.
 */
#ifndef ACADOS_SOLVER_CRAZYFLIE_H_
#define ACADOS_SOLVER_CRAZYFLIE_H_

#ifdef __cplusplus
extern "C" {
#endif

/* ---- dimensions fixed by generate_c_code.py:41-46 (acados_mpc.cpp:96-104 re-#defines them,
 *      so no macros are exported here) */
enum { CRAZYFLIE_N = 50, CRAZYFLIE_NX = 13, CRAZYFLIE_NU = 4, CRAZYFLIE_NY = 17, CRAZYFLIE_NYN = 13 };

/* ---- opaque-ish acados types the caller declares globals of (acados_mpc.cpp:76-84) */
typedef struct ocp_nlp_dims { int N; int nx; int nu; int ny; int ny_e; } ocp_nlp_dims;
typedef struct ocp_nlp_config { int N; } ocp_nlp_config;
typedef struct ocp_nlp_plan { int nlp_solver; } ocp_nlp_plan;
typedef struct ocp_nlp_in { void *priv; } ocp_nlp_in;
typedef struct ocp_nlp_solver { void *priv; } ocp_nlp_solver;
typedef struct ocp_nlp_out {
    double inf_norm_res; /* max-norm residual of the last QP; read at acados_mpc.cpp:615 */
    double total_time;   /* wall time of the last acados_solve() [s];          read at acados_mpc.cpp:616 */
    int sqp_iter;        /* always 1 (SQP_RTI)                                                            */
    int qp_iter;         /* interior-point iterations of the last QP                                      */
    void *priv;
} ocp_nlp_out;
typedef struct external_function_param_casadi { void *priv; } external_function_param_casadi;

/* The CALLER defines these (acados_mpc.cpp:76-84); acados_create() fills them. */
extern ocp_nlp_in *nlp_in;
extern ocp_nlp_out *nlp_out;
extern ocp_nlp_solver *nlp_solver;
extern void *nlp_opts;
extern ocp_nlp_plan *nlp_solver_plan;
extern ocp_nlp_config *nlp_config;
extern ocp_nlp_dims *nlp_dims;
extern external_function_param_casadi *forw_vde_casadi;

/* ---- lifecycle: acados_mpc.cpp:225 / :611 / :418 */
int acados_create(void);
int acados_solve(void);
int acados_free(void);

/* ---- setters (copy-in), acados_mpc.cpp:581-582, 590-594, 599-601, 606-607
 *   constraints: stage 0 "lbx"/"ubx" (13 doubles; the solver pins x0 = lbx and requires
 *                ubx == lbx at solve time), "lbu"/"ubu" (4 doubles, stage 0..N-1): stored PER STAGE and
 *                PER INPUT, returns 0; the next acados_solve() applies what is stored -- one box for
 *                all inputs and stages through cfnmpc_set_box when the values are uniform, the per-stage
 *                boxes through cfnmpc_set_box_stages otherwise (lb == ub pins an input: the reference's
 *                FIXED_U0 variant, which fixes stage 0 to the input in flight, acados_mpc.cpp:605-608,
 *                runs as written; tests/test_gpu_box_stages.py) -- and returns 1 without solving only
 *                if they are inadmissible (lb > ub or NaN somewhere)
 *   cost:        "yref" (17 doubles for stage < N, 13 for stage N); "W" (17x17 resp. 13x13,
 *                diagonal read from either major order; applies to every stage; state weights
 *                >= 0, input weights > 0, checked as a whole before anything is stored) */
int ocp_nlp_constraints_model_set(ocp_nlp_config *config, ocp_nlp_dims *dims, ocp_nlp_in *in, int stage,
                                  const char *field, void *value);
int ocp_nlp_cost_model_set(ocp_nlp_config *config, ocp_nlp_dims *dims, ocp_nlp_in *in, int stage,
                           const char *field, void *value);
/* ---- getter (copy-out), acados_mpc.cpp:619-625, 681-682: "x" (13) stage 0..N, "u" (4) stage 0..N-1 */
void ocp_nlp_out_get(ocp_nlp_config *config, ocp_nlp_dims *dims, ocp_nlp_out *out, int stage,
                     const char *field, void *value);

/* ---- extension (not in acados): choose the initial iterate after acados_create():
 *      0 = acados default (x_k = [0,0,0,1,0..], u_k = 0), 1 = x_k = current lbx, u_k = hover. */
int acados_cfnmpc_init_iterate(int mode);

#ifdef __cplusplus
}
#endif
#endif /* ACADOS_SOLVER_CRAZYFLIE_H_ */
