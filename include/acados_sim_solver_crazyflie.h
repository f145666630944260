/* acados_sim_solver_crazyflie.h -- drop-in replacement of the GENERATED acados sim-solver header
 * that crazyflie_controller/src/acados_estimator.cpp:74-76 includes, plus the acados_c sim
 * entry points the estimator calls (acados_estimator.cpp:237, 573-593).  Implemented by
 * libacados_solver_crazyflie.so over cfnmpc_sim() (include/cfnmpc.h) with batch = 1.
 *
 * The predictor integrates the model ODE from x over T with the input u held constant
 * (one explicit RK4 integration; SURVEY App. D-8: 4 sub-steps).  "T" is the delay to
 * compensate (0.06 s in crazyflie_controller/launch/acados_predictor.launch:62).
 */
#ifndef ACADOS_SIM_SOLVER_CRAZYFLIE_H_
#define ACADOS_SIM_SOLVER_CRAZYFLIE_H_

#ifdef __cplusplus
extern "C" {
#endif

/* ns = stages of the explicit Runge-Kutta scheme (4: classic RK4, the only one offered);
 * num_steps = integration steps over T (default 4, i.e. 15 ms steps for T = 60 ms) */
typedef struct sim_config { int ns; int num_steps; } sim_config;
typedef struct sim_in { double T; double x[13]; double u[4]; } sim_in;
typedef struct sim_out { double xn[13]; double total_time; } sim_out;

/* globals of the generated sim solver, used at acados_estimator.cpp:573-593; defined by the
 * library (the estimator only declares the casadi pointer, acados_estimator.cpp:97) */
extern sim_config *crazyflie_sim_config;
extern void *crazyflie_sim_dims;
extern sim_in *crazyflie_sim_in;
extern sim_out *crazyflie_sim_out;

int crazyflie_acados_sim_create(void); /* acados_estimator.cpp:237 */
int crazyflie_acados_sim_solve(void);  /* acados_estimator.cpp:589 */
int crazyflie_acados_sim_free(void);

/* field in {"T" (1 double), "x" (13), "u" (4)}: acados_estimator.cpp:573,578,586 */
int sim_in_set(void *config, void *dims, sim_in *in, const char *field, void *value);
/* field "xn" (13 doubles): acados_estimator.cpp:593 */
int sim_out_get(void *config, void *dims, sim_out *out, const char *field, void *value);

#ifdef __cplusplus
}
#endif
#endif /* ACADOS_SIM_SOLVER_CRAZYFLIE_H_ */
