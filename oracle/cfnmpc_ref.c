/* cfnmpc_ref.c -- plain-C (FP64, dense) CPU restatement of the Crazyflie SQP-RTI hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/cfnmpc_oracle.py header): the
 * reference's arithmetic is acados/HPIPM/BLASFEO, an empty un-pinned submodule
 * (/root/reference/.gitmodules:7-10); this file restates the mathematics the reference pins
 *   - ODE + constants ........ crazyflie_controller/scripts/crazyflie_full_model/export_ode_model.py:33-102
 *   - OCP definition ......... crazyflie_controller/scripts/crazyflie_full_model/generate_c_code.py:41-146
 *   - per-step protocol ...... crazyflie_controller/src/acados_mpc.cpp:581-625
 *   - predictor .............. crazyflie_controller/src/acados_estimator.cpp:573-593
 * and is validated against oracle/cfnmpc_oracle.py (sympy Jacobian, dense QP solver) and the
 * reference's own traj/smooth_step.txt (tests/test_oracle_golden.py).
 *
 * It deliberately uses DENSE 13x13 / 13x4 algebra (the HIP kernels exploit the sparsity of
 * d Phi / d x), so that the two implementations share the algorithm but
 * not the code.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load the library built from this file; the product (libcfnmpc.so) never links it.
 *
 * Build: make -C oracle   ->  oracle/libcfnmpc_oracle.so
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NX 13
#define NU 4
#define NY 17

/* export_ode_model.py:34-42 */
static const double G0 = 9.8066, MQ = 33e-3, IXX = 1.395e-5, IYY = 1.395e-5, IZZ = 2.173e-5,
                    CD = 7.9379e-06, CT = 3.25e-4, ARM = 65e-3 / 2;

typedef struct {
    int N;            /* horizon length (generate_c_code.py:42: 50)            */
    double dt;        /* shooting interval (Tf/N = 0.015)                      */
    double W[NY];     /* diag of stage weight (generate_c_code.py:63-84)       */
    double WN[NX];    /* diag of terminal weight (generate_c_code.py:109)      */
    double u_min, u_max; /* generate_c_code.py:133-134                         */
    double tol;       /* IPM: max-norm tolerance on all QP residuals           */
    int max_iter;     /* IPM: iteration cap                                    */
    double tau;       /* IPM: fraction to the boundary                         */
    double thr0;      /* IPM: slack floor of the starting point                */
    double lam0_min;  /* IPM: floor of the starting complementarity            */
    double mu0_scale; /* IPM: starting complementarity = max(lam0_min, mu0_scale * max. violation) */
    int active_set;   /* 1: primal-dual active-set solves first (the engine's cfnmpc_opts.active_set;
                         numpy twin pdas_dense), interior point only if the set does not settle in
                         12 solves; 0 (default): interior point only                              */
    double clip_viol; /* IPM: clipped start when the unconstrained minimiser leaves the box by more than this
                         many box widths (cfnmpc_opts.ipm_clip_viol; 0: never)                   */
    double clip_margin; /* ... clipped to this fraction of the box width inside the bounds       */
    double as_skip_viol; /* active_set: beyond this many box widths the active-set solves are skipped
                         (cfnmpc_opts.as_skip_viol; 0: never)                                    */
    int as_warm;      /* active_set: warm start of the classification from the instance's previous RTI step
                         (cfnmpc_opts.as_warm; needs the per-instance state of cfo_rti_step_w)     */
} cfo_opts;

void cfo_default_opts(cfo_opts *o) {
    static const double W[NY] = {120.0, 100.0, 100.0, 1e-3, 1e-3, 1e-3, 1e-3, 0.7, 1.0, 4.0,
                                 1e-5,  1e-5,  10.0,  0.06, 0.06, 0.06, 0.06};
    o->N = 50;
    o->dt = 0.75 / 50;
    for (int i = 0; i < NY; i++) o->W[i] = W[i];
    for (int i = 0; i < NX; i++) o->WN[i] = 50.0 * W[i];
    o->u_min = 0.0;
    o->u_max = 22.0;
    o->tol = 1e-8;
    o->max_iter = 50;
    o->tau = 0.995;
    o->thr0 = 1.0;
    o->lam0_min = 1e-2;
    o->mu0_scale = 0.1;
    o->active_set = 0;
    o->clip_viol = 2.0;
    o->clip_margin = 0.05;
    o->as_skip_viol = 4.0;
    o->as_warm = 0;
}

/* ---------------------------------------------------------------- dynamics */
void cfo_f(const double *x, const double *u, double *dx) {
    const double q1 = x[3], q2 = x[4], q3 = x[5], q4 = x[6];
    const double vbx = x[7], vby = x[8], vbz = x[9];
    const double wx = x[10], wy = x[11], wz = x[12];
    const double w1 = u[0], w2 = u[1], w3 = u[2], w4 = u[3];
    dx[0] = vbx * (2 * q1 * q1 + 2 * q2 * q2 - 1) - vby * (2 * q1 * q4 - 2 * q2 * q3) + vbz * (2 * q1 * q3 + 2 * q2 * q4);
    dx[1] = vby * (2 * q1 * q1 + 2 * q3 * q3 - 1) + vbx * (2 * q1 * q4 + 2 * q2 * q3) - vbz * (2 * q1 * q2 - 2 * q3 * q4);
    dx[2] = vbz * (2 * q1 * q1 + 2 * q4 * q4 - 1) - vbx * (2 * q1 * q3 - 2 * q2 * q4) + vby * (2 * q1 * q2 + 2 * q3 * q4);
    dx[3] = -(q2 * wx) / 2 - (q3 * wy) / 2 - (q4 * wz) / 2;
    dx[4] = (q1 * wx) / 2 - (q4 * wy) / 2 + (q3 * wz) / 2;
    dx[5] = (q4 * wx) / 2 + (q1 * wy) / 2 - (q2 * wz) / 2;
    dx[6] = (q2 * wy) / 2 - (q3 * wx) / 2 + (q1 * wz) / 2;
    dx[7] = vby * wz - vbz * wy + G0 * (2 * q1 * q3 - 2 * q2 * q4);
    dx[8] = vbz * wx - vbx * wz - G0 * (2 * q1 * q2 + 2 * q3 * q4);
    dx[9] = vbx * wy - vby * wx - G0 * (2 * q1 * q1 + 2 * q4 * q4 - 1) + (CT * (w1 * w1 + w2 * w2 + w3 * w3 + w4 * w4)) / MQ;
    dx[10] = -(CT * ARM * (w1 * w1 + w2 * w2 - w3 * w3 - w4 * w4) - IYY * wy * wz + IZZ * wy * wz) / IXX;
    dx[11] = -(CT * ARM * (w1 * w1 - w2 * w2 - w3 * w3 + w4 * w4) + IXX * wx * wz - IZZ * wx * wz) / IYY;
    dx[12] = -(CD * (w1 * w1 - w2 * w2 + w3 * w3 - w4 * w4) - IXX * wx * wy + IYY * wx * wy) / IZZ;
}

/* J[13][17] = [df/dx | df/du], row-major, hand-derived from export_ode_model.py:85-97 */
void cfo_jac(const double *x, const double *u, double *J) {
    const double q1 = x[3], q2 = x[4], q3 = x[5], q4 = x[6];
    const double vx = x[7], vy = x[8], vz = x[9];
    const double wx = x[10], wy = x[11], wz = x[12];
    memset(J, 0, sizeof(double) * NX * NY);
#define JJ(i, j) J[(i)*NY + (j)]
    /* position rows */
    JJ(0, 3) = 4 * q1 * vx - 2 * q4 * vy + 2 * q3 * vz;
    JJ(0, 4) = 4 * q2 * vx + 2 * q3 * vy + 2 * q4 * vz;
    JJ(0, 5) = 2 * q2 * vy + 2 * q1 * vz;
    JJ(0, 6) = -2 * q1 * vy + 2 * q2 * vz;
    JJ(0, 7) = 2 * q1 * q1 + 2 * q2 * q2 - 1;
    JJ(0, 8) = -(2 * q1 * q4 - 2 * q2 * q3);
    JJ(0, 9) = 2 * q1 * q3 + 2 * q2 * q4;
    JJ(1, 3) = 4 * q1 * vy + 2 * q4 * vx - 2 * q2 * vz;
    JJ(1, 4) = 2 * q3 * vx - 2 * q1 * vz;
    JJ(1, 5) = 4 * q3 * vy + 2 * q2 * vx + 2 * q4 * vz;
    JJ(1, 6) = 2 * q1 * vx + 2 * q3 * vz;
    JJ(1, 7) = 2 * q1 * q4 + 2 * q2 * q3;
    JJ(1, 8) = 2 * q1 * q1 + 2 * q3 * q3 - 1;
    JJ(1, 9) = -(2 * q1 * q2 - 2 * q3 * q4);
    JJ(2, 3) = 4 * q1 * vz - 2 * q3 * vx + 2 * q2 * vy;
    JJ(2, 4) = 2 * q4 * vx + 2 * q1 * vy;
    JJ(2, 5) = -2 * q1 * vx + 2 * q4 * vy;
    JJ(2, 6) = 4 * q4 * vz + 2 * q2 * vx + 2 * q3 * vy;
    JJ(2, 7) = -(2 * q1 * q3 - 2 * q2 * q4);
    JJ(2, 8) = 2 * q1 * q2 + 2 * q3 * q4;
    JJ(2, 9) = 2 * q1 * q1 + 2 * q4 * q4 - 1;
    /* quaternion rows */
    JJ(3, 4) = -wx / 2; JJ(3, 5) = -wy / 2; JJ(3, 6) = -wz / 2;
    JJ(3, 10) = -q2 / 2; JJ(3, 11) = -q3 / 2; JJ(3, 12) = -q4 / 2;
    JJ(4, 3) = wx / 2; JJ(4, 5) = wz / 2; JJ(4, 6) = -wy / 2;
    JJ(4, 10) = q1 / 2; JJ(4, 11) = -q4 / 2; JJ(4, 12) = q3 / 2;
    JJ(5, 3) = wy / 2; JJ(5, 4) = -wz / 2; JJ(5, 6) = wx / 2;
    JJ(5, 10) = q4 / 2; JJ(5, 11) = q1 / 2; JJ(5, 12) = -q2 / 2;
    JJ(6, 3) = wz / 2; JJ(6, 4) = wy / 2; JJ(6, 5) = -wx / 2;
    JJ(6, 10) = -q3 / 2; JJ(6, 11) = q2 / 2; JJ(6, 12) = q1 / 2;
    /* body velocity rows */
    JJ(7, 3) = 2 * G0 * q3; JJ(7, 4) = -2 * G0 * q4; JJ(7, 5) = 2 * G0 * q1; JJ(7, 6) = -2 * G0 * q2;
    JJ(7, 8) = wz; JJ(7, 9) = -wy; JJ(7, 11) = -vz; JJ(7, 12) = vy;
    JJ(8, 3) = -2 * G0 * q2; JJ(8, 4) = -2 * G0 * q1; JJ(8, 5) = -2 * G0 * q4; JJ(8, 6) = -2 * G0 * q3;
    JJ(8, 7) = -wz; JJ(8, 9) = wx; JJ(8, 10) = vz; JJ(8, 12) = -vx;
    JJ(9, 3) = -4 * G0 * q1; JJ(9, 6) = -4 * G0 * q4;
    JJ(9, 7) = wy; JJ(9, 8) = -wx; JJ(9, 10) = -vy; JJ(9, 11) = vx;
    for (int i = 0; i < NU; i++) JJ(9, NX + i) = 2 * CT * u[i] / MQ;
    /* body rate rows */
    JJ(10, 11) = -(IZZ - IYY) * wz / IXX; JJ(10, 12) = -(IZZ - IYY) * wy / IXX;
    JJ(11, 10) = -(IXX - IZZ) * wz / IYY; JJ(11, 12) = -(IXX - IZZ) * wx / IYY;
    JJ(12, 10) = -(IYY - IXX) * wy / IZZ; JJ(12, 11) = -(IYY - IXX) * wx / IZZ;
    {
        const double ka = -2 * CT * ARM / IXX, kb = -2 * CT * ARM / IYY, kc = -2 * CD / IZZ;
        JJ(10, 13) = ka * u[0]; JJ(10, 14) = ka * u[1]; JJ(10, 15) = -ka * u[2]; JJ(10, 16) = -ka * u[3];
        JJ(11, 13) = kb * u[0]; JJ(11, 14) = -kb * u[1]; JJ(11, 15) = -kb * u[2]; JJ(11, 16) = kb * u[3];
        JJ(12, 13) = kc * u[0]; JJ(12, 14) = -kc * u[1]; JJ(12, 15) = kc * u[2]; JJ(12, 16) = -kc * u[3];
    }
#undef JJ
}

/* classic RK4, `steps` sub-steps over dt */
void cfo_rk4(const double *x, const double *u, double dt, int steps, double *xn) {
    double xc[NX], k1[NX], k2[NX], k3[NX], k4[NX], xt[NX];
    const double h = dt / steps;
    memcpy(xc, x, sizeof xc);
    for (int s = 0; s < steps; s++) {
        cfo_f(xc, u, k1);
        for (int i = 0; i < NX; i++) xt[i] = xc[i] + 0.5 * h * k1[i];
        cfo_f(xt, u, k2);
        for (int i = 0; i < NX; i++) xt[i] = xc[i] + 0.5 * h * k2[i];
        cfo_f(xt, u, k3);
        for (int i = 0; i < NX; i++) xt[i] = xc[i] + h * k3[i];
        cfo_f(xt, u, k4);
        for (int i = 0; i < NX; i++) xc[i] += (h / 6.0) * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
    }
    memcpy(xn, xc, sizeof xc);
}

/* one VDE stage: k = f(xs,u); K = J_x(xs,u) S + [0 J_u]; S is 13x17 row-major */
static void vde_stage(const double *xs, const double *u, const double *S, double *k, double *K) {
    double J[NX * NY];
    cfo_f(xs, u, k);
    cfo_jac(xs, u, J);
    for (int i = 0; i < NX; i++)
        for (int j = 0; j < NY; j++) {
            double acc = (j >= NX) ? J[i * NY + j] : 0.0;
            for (int l = 0; l < NX; l++) acc += J[i * NY + l] * S[l * NY + j];
            K[i * NY + j] = acc;
        }
}

/* RK4 with forward sensitivities: phi[13], A[13x13], B[13x4] (row-major) */
void cfo_rk4_sens(const double *x, const double *u, double dt, double *phi, double *A, double *B) {
    double S0[NX * NY], St[NX * NY], xt[NX];
    double k1[NX], k2[NX], k3[NX], k4[NX];
    double K1[NX * NY], K2[NX * NY], K3[NX * NY], K4[NX * NY];
    memset(S0, 0, sizeof S0);
    for (int i = 0; i < NX; i++) S0[i * NY + i] = 1.0;
    vde_stage(x, u, S0, k1, K1);
    for (int i = 0; i < NX; i++) xt[i] = x[i] + 0.5 * dt * k1[i];
    for (int i = 0; i < NX * NY; i++) St[i] = S0[i] + 0.5 * dt * K1[i];
    vde_stage(xt, u, St, k2, K2);
    for (int i = 0; i < NX; i++) xt[i] = x[i] + 0.5 * dt * k2[i];
    for (int i = 0; i < NX * NY; i++) St[i] = S0[i] + 0.5 * dt * K2[i];
    vde_stage(xt, u, St, k3, K3);
    for (int i = 0; i < NX; i++) xt[i] = x[i] + dt * k3[i];
    for (int i = 0; i < NX * NY; i++) St[i] = S0[i] + dt * K3[i];
    vde_stage(xt, u, St, k4, K4);
    for (int i = 0; i < NX; i++) phi[i] = x[i] + (dt / 6.0) * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
    for (int i = 0; i < NX; i++)
        for (int j = 0; j < NY; j++) {
            double s = S0[i * NY + j] + (dt / 6.0) * (K1[i * NY + j] + 2 * K2[i * NY + j] + 2 * K3[i * NY + j] + K4[i * NY + j]);
            if (j < NX) A[i * NX + j] = s; else B[i * NU + (j - NX)] = s;
        }
}

/* ---------------------------------------------------------------- stage QP */
typedef struct {
    int N;
    double *A, *B, *b, *q, *r, *lb, *ub; /* [N][169] [N][52] [N][13] [N+1][13] [N][4] [N][4] [N][4] */
    double dx0[NX];
    const double *Qd, *Rd, *QNd;
    /* factorisation + work */
    double *K, *Sinv, *d;              /* [N][52] [N][16] [N][4] */
    double *v, *tl, *tu, *ll, *lu, *rg; /* [N][4] each */
    double *dva, *dvc, *x;              /* [N][4] [N][4] [(N+1)][13] */
    /* warm start of the active set (cfo_opts.as_warm): this instance's state, or NULL -- the classes its last SETTLED
     * active-set solve ended with ([N][4]: 0 free, 1 lower, 2 upper) and whether its previous RTI step ended that way */
    unsigned char *wcls;
    int *wvalid;
} qp_t;

static size_t qp_doubles(int N) {
    return (size_t)N * (169 + 52 + 13 + 4 + 4 + 4 + 52 + 16 + 4 + 6 * 4 + 2 * 4) + 2 * (size_t)(N + 1) * 13;
}

static void qp_carve(qp_t *qp, int N, double *m) {
    qp->N = N;
    qp->wcls = NULL;
    qp->wvalid = NULL;
    qp->A = m; m += (size_t)N * 169;
    qp->B = m; m += (size_t)N * 52;
    qp->b = m; m += (size_t)N * 13;
    qp->q = m; m += (size_t)(N + 1) * 13;
    qp->r = m; m += (size_t)N * 4;
    qp->lb = m; m += (size_t)N * 4;
    qp->ub = m; m += (size_t)N * 4;
    qp->K = m; m += (size_t)N * 52;
    qp->Sinv = m; m += (size_t)N * 16;
    qp->d = m; m += (size_t)N * 4;
    qp->v = m; m += (size_t)N * 4;
    qp->tl = m; m += (size_t)N * 4;
    qp->tu = m; m += (size_t)N * 4;
    qp->ll = m; m += (size_t)N * 4;
    qp->lu = m; m += (size_t)N * 4;
    qp->rg = m; m += (size_t)N * 4;
    qp->dva = m; m += (size_t)N * 4;
    qp->dvc = m; m += (size_t)N * 4;
    qp->x = m;
}

/* symmetric 4x4 inverse through Cholesky; returns 0 ok / 1 not positive definite */
static int spd4_inv(const double *S, double *Si) {
    double L[16] = {0}, Li[16] = {0};
    for (int j = 0; j < 4; j++) {
        double s = S[j * 4 + j];
        for (int k = 0; k < j; k++) s -= L[j * 4 + k] * L[j * 4 + k];
        if (!(s > 0.0)) return 1;
        L[j * 4 + j] = sqrt(s);
        for (int i = j + 1; i < 4; i++) {
            double t = S[i * 4 + j];
            for (int k = 0; k < j; k++) t -= L[i * 4 + k] * L[j * 4 + k];
            L[i * 4 + j] = t / L[j * 4 + j];
        }
    }
    for (int j = 0; j < 4; j++) { /* Li = inv(L), lower */
        Li[j * 4 + j] = 1.0 / L[j * 4 + j];
        for (int i = j + 1; i < 4; i++) {
            double t = 0;
            for (int k = j; k < i; k++) t -= L[i * 4 + k] * Li[k * 4 + j];
            Li[i * 4 + j] = t / L[i * 4 + i];
        }
    }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double t = 0;
            for (int k = (i > j ? i : j); k < 4; k++) t += Li[k * 4 + i] * Li[k * 4 + j];
            Si[i * 4 + j] = t;
        }
    return 0;
}

/* backward sweep: Riccati factorisation with input Hessian diag Rhat[k] and gradient g[k];
 * absolute != 0 -> affine terms q, b of the QP are included (start solve). */
static int riccati_factor_aff(qp_t *qp, const double *Rhat, const double *g, const double *bb, const double *qq);
static int riccati_factor(qp_t *qp, const double *Rhat, const double *g, int absolute) {
    return riccati_factor_aff(qp, Rhat, g, absolute ? qp->b : NULL, absolute ? qp->q : NULL);
}
/* bb [N][13] / qq [N+1][13]: affine terms of the dynamics / of the state cost (NULL = zero) */
static int riccati_factor_aff(qp_t *qp, const double *Rhat, const double *g, const double *bb, const double *qq) {
    const int N = qp->N;
    double P[169], p[NX], PA[169], PB[52], S[16], G[52], hb[NX], rho[4], Pn[169], pn[NX];
    memset(P, 0, sizeof P);
    for (int i = 0; i < NX; i++) { P[i * NX + i] = qp->QNd[i]; p[i] = qq ? qq[N * NX + i] : 0.0; }
    for (int k = N - 1; k >= 0; k--) {
        const double *A = qp->A + (size_t)k * 169, *B = qp->B + (size_t)k * 52, *b = bb ? bb + (size_t)k * NX : NULL;
        double *K = qp->K + (size_t)k * 52, *Si = qp->Sinv + (size_t)k * 16, *d = qp->d + (size_t)k * 4;
        for (int i = 0; i < NX; i++) {
            for (int j = 0; j < NX; j++) { double s = 0; for (int l = 0; l < NX; l++) s += P[i * NX + l] * A[l * NX + j]; PA[i * NX + j] = s; }
            for (int j = 0; j < NU; j++) { double s = 0; for (int l = 0; l < NX; l++) s += P[i * NX + l] * B[l * NU + j]; PB[i * NU + j] = s; }
        }
        for (int i = 0; i < NU; i++) {
            for (int j = 0; j < NU; j++) { double s = (i == j) ? Rhat[k * 4 + i] : 0.0; for (int l = 0; l < NX; l++) s += B[l * NU + i] * PB[l * NU + j]; S[i * 4 + j] = s; }
            for (int j = 0; j < NX; j++) { double s = 0; for (int l = 0; l < NX; l++) s += B[l * NU + i] * PA[l * NX + j]; G[i * NX + j] = s; }
        }
        for (int i = 0; i < NX; i++) {
            double s = p[i];
            if (b) for (int l = 0; l < NX; l++) s += P[i * NX + l] * b[l];
            hb[i] = s;
        }
        for (int i = 0; i < NU; i++) { double s = g[k * 4 + i]; for (int l = 0; l < NX; l++) s += B[l * NU + i] * hb[l]; rho[i] = s; }
        if (spd4_inv(S, Si)) return 1;
        for (int i = 0; i < NU; i++) {
            for (int j = 0; j < NX; j++) { double s = 0; for (int l = 0; l < NU; l++) s += Si[i * 4 + l] * G[l * NX + j]; K[i * NX + j] = s; }
            double s = 0; for (int l = 0; l < NU; l++) s += Si[i * 4 + l] * rho[l]; d[i] = s;
        }
        for (int i = 0; i < NX; i++)
            for (int j = 0; j < NX; j++) {
                double s = (i == j) ? qp->Qd[i] : 0.0;
                for (int l = 0; l < NX; l++) s += A[l * NX + i] * PA[l * NX + j];
                for (int l = 0; l < NU; l++) s -= G[l * NX + i] * K[l * NX + j];
                Pn[i * NX + j] = s;
            }
        for (int i = 0; i < NX; i++) {
            double s = qq ? qq[k * NX + i] : 0.0;
            for (int l = 0; l < NX; l++) s += A[l * NX + i] * hb[l];
            for (int l = 0; l < NU; l++) s -= K[l * NX + i] * rho[l];
            pn[i] = s;
        }
        for (int i = 0; i < NX; i++) { p[i] = pn[i]; for (int j = 0; j < NX; j++) P[i * NX + j] = 0.5 * (Pn[i * NX + j] + Pn[j * NX + i]); }
    }
    return 0;
}

/* re-use the factorisation for a new input-row right-hand side g (homogeneous problem) */
static void riccati_resolve(qp_t *qp, const double *g) {
    const int N = qp->N;
    double p[NX] = {0}, pn[NX], rho[4];
    for (int k = N - 1; k >= 0; k--) {
        const double *A = qp->A + (size_t)k * 169, *B = qp->B + (size_t)k * 52;
        const double *K = qp->K + (size_t)k * 52, *Si = qp->Sinv + (size_t)k * 16;
        double *d = qp->d + (size_t)k * 4;
        for (int i = 0; i < NU; i++) { double s = g[k * 4 + i]; for (int l = 0; l < NX; l++) s += B[l * NU + i] * p[l]; rho[i] = s; }
        for (int i = 0; i < NU; i++) { double s = 0; for (int l = 0; l < NU; l++) s += Si[i * 4 + l] * rho[l]; d[i] = s; }
        for (int i = 0; i < NX; i++) {
            double s = 0;
            for (int l = 0; l < NX; l++) s += A[l * NX + i] * p[l];
            for (int l = 0; l < NU; l++) s -= K[l * NX + i] * rho[l];
            pn[i] = s;
        }
        memcpy(p, pn, sizeof p);
    }
}

/* forward sweep: v_k = -K x - d ; x+ = A x + B v (+ b).  Writes v[N][4] and x[(N+1)][13] */
static void riccati_forward(const qp_t *qp, int absolute, double *v, double *xs) {
    const int N = qp->N;
    double x[NX], xn[NX];
    for (int i = 0; i < NX; i++) x[i] = absolute ? qp->dx0[i] : 0.0;
    memcpy(xs, x, sizeof x);
    for (int k = 0; k < N; k++) {
        const double *A = qp->A + (size_t)k * 169, *B = qp->B + (size_t)k * 52, *b = qp->b + (size_t)k * NX;
        const double *K = qp->K + (size_t)k * 52, *d = qp->d + (size_t)k * 4;
        double *vk = v + (size_t)k * 4;
        for (int i = 0; i < NU; i++) { double s = -d[i]; for (int l = 0; l < NX; l++) s -= K[i * NX + l] * x[l]; vk[i] = s; }
        for (int i = 0; i < NX; i++) {
            double s = absolute ? b[i] : 0.0;
            for (int l = 0; l < NX; l++) s += A[i * NX + l] * x[l];
            for (int l = 0; l < NU; l++) s += B[i * NU + l] * vk[l];
            xn[i] = s;
        }
        memcpy(x, xn, sizeof x);
        memcpy(xs + (size_t)(k + 1) * NX, x, sizeof x);
    }
}

static void rollout(const qp_t *qp, const double *v, double *xs) {
    const int N = qp->N;
    double x[NX], xn[NX];
    memcpy(x, qp->dx0, sizeof x);
    memcpy(xs, x, sizeof x);
    for (int k = 0; k < N; k++) {
        const double *A = qp->A + (size_t)k * 169, *B = qp->B + (size_t)k * 52, *b = qp->b + (size_t)k * NX;
        for (int i = 0; i < NX; i++) {
            double s = b[i];
            for (int l = 0; l < NX; l++) s += A[i * NX + l] * x[l];
            for (int l = 0; l < NU; l++) s += B[i * NU + l] * v[k * 4 + l];
            xn[i] = s;
        }
        memcpy(x, xn, sizeof x);
        memcpy(xs + (size_t)(k + 1) * NX, x, sizeof x);
    }
}

static double steplen(int n, const double *z, const double *dz, double a) {
    for (int i = 0; i < n; i++)
        if (dz[i] < 0.0) { double t = -z[i] / dz[i]; if (t < a) a = t; }
    return a;
}

/* Primal-dual active-set solves in delta form around the unconstrained minimiser qp->v
 * (numpy twin pdas_dense; the engine's k_as): classify every input, solve
 * the homogeneous LQ problem with the active inputs fixed at c = bound - v0 (1e30 on their diagonal
 * of R^, b_eff = B c in the affine recursion), forward sweep, costate sweep with multipliers
 * R c + B'pi and re-classification; a stationary classification is the KKT system.
 * Returns the number of solves (> 0) with qp->v = solution, or 0 if the set did not settle. */
static int as_solve(qp_t *qp) {
    const int N = qp->N, n = N * NU;
    double *v0 = qp->v, *du = qp->dva;
    double *Rhat = (double *)malloc(sizeof(double) * ((size_t)n * 3 + (size_t)N * NX + (size_t)(N + 1) * NX));
    double *g = Rhat + n, *c = g + n, *beff = c + n, *dx = beff + (size_t)N * NX;
    int *cls = (int *)malloc(sizeof(int) * n);
    int solves = 0, done = 0;
    /* WARM: the union of the previous step's final set and today's violations (the engine's qp_wave does the same; any
     * start ends in a stationary classification = the exact solution, only the number of solves differs) */
    const int warm = qp->wvalid && *qp->wvalid;
    for (int i = 0; i < n; i++) {
        cls[i] = v0[i] < qp->lb[i] ? 1 : (v0[i] > qp->ub[i] ? 2 : 0);
        if (warm && cls[i] == 0 && qp->wcls[i]) cls[i] = qp->wcls[i];
        c[i] = cls[i] == 1 ? qp->lb[i] - v0[i] : (cls[i] == 2 ? qp->ub[i] - v0[i] : 0.0);
        g[i] = 0.0;
    }
    while (solves < 12 && !done) {
        solves++;
        for (int k = 0; k < N; k++) {
            const double *B = qp->B + (size_t)k * 52;
            for (int a = 0; a < NU; a++) Rhat[k * 4 + a] = cls[k * 4 + a] ? 1e30 * fmax(1.0, qp->Rd[a]) : qp->Rd[a];
            for (int i = 0; i < NX; i++) {
                double s = 0.0;
                for (int a = 0; a < NU; a++) s += B[i * NU + a] * c[k * 4 + a];
                beff[k * NX + i] = s;
            }
        }
        if (riccati_factor_aff(qp, Rhat, g, beff, NULL)) { solves = 0; break; }
        {   /* forward: free inputs from the feedback law, fixed ones = c */
            double x[NX] = {0}, xn[NX];
            memcpy(dx, x, sizeof x);
            for (int k = 0; k < N; k++) {
                const double *A = qp->A + (size_t)k * 169, *B = qp->B + (size_t)k * 52;
                const double *K = qp->K + (size_t)k * 52, *d = qp->d + (size_t)k * 4;
                for (int a = 0; a < NU; a++) {
                    double s = -d[a];
                    for (int l = 0; l < NX; l++) s -= K[a * NX + l] * x[l];
                    du[k * 4 + a] = cls[k * 4 + a] ? c[k * 4 + a] : s;
                }
                for (int i = 0; i < NX; i++) {
                    double s = 0.0;
                    for (int l = 0; l < NX; l++) s += A[i * NX + l] * x[l];
                    for (int a = 0; a < NU; a++) s += B[i * NU + a] * du[k * 4 + a];
                    xn[i] = s;
                }
                memcpy(x, xn, sizeof x);
                memcpy(dx + (size_t)(k + 1) * NX, x, sizeof x);
            }
        }
        {   /* costate: pi_N = QN dx_N ; multipliers and new classification ; pi_k = Q dx_k + A'pi */
            double pi[NX], pn[NX];
            int changed = 0;
            for (int i = 0; i < NX; i++) pi[i] = qp->QNd[i] * dx[(size_t)N * NX + i];
            for (int k = N - 1; k >= 0; k--) {
                const double *A = qp->A + (size_t)k * 169, *B = qp->B + (size_t)k * 52;
                for (int a = 0; a < NU; a++) {
                    const int i = k * 4 + a;
                    double grad = qp->Rd[a] * c[i];
                    for (int l = 0; l < NX; l++) grad += B[l * NU + a] * pi[l];
                    const double vn = v0[i] + du[i];
                    int nc;
                    if (cls[i] == 0) nc = vn < qp->lb[i] ? 1 : (vn > qp->ub[i] ? 2 : 0);
                    else if (cls[i] == 1) nc = grad > 0.0 ? 1 : 0;
                    else nc = grad < 0.0 ? 2 : 0;
                    if (nc != cls[i]) changed = 1;
                    cls[i] = nc;
                    c[i] = nc == 1 ? qp->lb[i] - v0[i] : (nc == 2 ? qp->ub[i] - v0[i] : 0.0);
                }
                for (int i = 0; i < NX; i++) {
                    double s = qp->Qd[i] * dx[(size_t)k * NX + i];
                    for (int l = 0; l < NX; l++) s += A[l * NX + i] * pi[l];
                    pn[i] = s;
                }
                memcpy(pi, pn, sizeof pi);
            }
            done = !changed;
        }
    }
    if (done) for (int i = 0; i < n; i++) v0[i] += du[i];
    if (qp->wvalid) {
        if (done) for (int i = 0; i < n; i++) qp->wcls[i] = (unsigned char)cls[i];
        *qp->wvalid = done;
    }
    free(cls);
    free(Rhat);
    return done ? solves : 0;
}

/* Mehrotra predictor-corrector, delta form (mirrors riccati_ipm in
 * cfnmpc_oracle.py).  On return qp->v holds du and qp->x holds dx.
 * status: 0 converged, 2 iteration cap, 4 factorisation failure / non-finite */
static int ipm_solve(qp_t *qp, const cfo_opts *o, int *iters_out, double *res_out) {
    const int N = qp->N, n = N * NU;
    const double nc = 2.0 * n;
    double *v = qp->v, *tl = qp->tl, *tu = qp->tu, *ll = qp->ll, *lu = qp->lu, *rg = qp->rg;
    double *dva = qp->dva, *dvc = qp->dvc;
    double *Rhat = (double *)malloc(sizeof(double) * n * 12);
    double *g = Rhat + n, *dtl = g + n, *dtu = dtl + n, *dll = dtu + n, *dlu = dll + n;
    double *rl = dlu + n, *ru = rl + n, *cl = ru + n, *cu = cl + n, *Dl = cu + n, *Du = Dl + n;
    int status = 2, it = 0;
    double res = 0.0;
    *iters_out = 0;
    for (int i = 0; i < n; i++) Rhat[i] = qp->Rd[i % NU];
    if (riccati_factor(qp, Rhat, qp->r, 1)) { free(Rhat); *res_out = NAN; return 4; }
    riccati_forward(qp, 1, v, qp->x);
    int feas = 1;
    double viol = 0.0;
    for (int i = 0; i < n; i++) {
        if (!(v[i] >= qp->lb[i] && v[i] <= qp->ub[i])) feas = 0;
        if (qp->lb[i] - v[i] > viol) viol = qp->lb[i] - v[i];
        if (v[i] - qp->ub[i] > viol) viol = v[i] - qp->ub[i];
    }
    if (feas) { if (qp->wvalid) *qp->wvalid = 0; free(Rhat); *res_out = 0.0; return 0; }
    if (!(viol == viol)) { if (qp->wvalid) *qp->wvalid = 0; free(Rhat); *res_out = NAN; return 4; }
    if (qp->wvalid && !(o->active_set && !(o->as_skip_viol > 0.0 && viol > o->as_skip_viol * (o->u_max - o->u_min)))) *qp->wvalid = 0;
    if (o->active_set && !(o->as_skip_viol > 0.0 && viol > o->as_skip_viol * (o->u_max - o->u_min))) {
        const int solves = as_solve(qp);
        if (solves > 0) {
            rollout(qp, v, qp->x);
            free(Rhat);
            *iters_out = solves;
            *res_out = 0.0;
            return 0;
        }
    }
    if (o->clip_viol > 0.0 && viol > o->clip_viol * (o->u_max - o->u_min)) {
        /* clipped start (riccati_ipm in cfnmpc_oracle.py): v inside the box, multipliers absorb the gradient
         * g = H (v - v0) of the condensed QP there (forward sweep for dx, backward costate sweep) */
        double *dxs = (double *)calloc((size_t)(N + 1) * NX, sizeof(double));
        double *dvc = dva, *gr = qp->dvc;   /* (both free before the first iteration) */
        for (int i = 0; i < n; i++) {
            const double w = qp->ub[i] - qp->lb[i];
            const double lo = qp->lb[i] + o->clip_margin * w, hi = qp->ub[i] - o->clip_margin * w;
            const double vc = v[i] < lo ? lo : (v[i] > hi ? hi : v[i]);
            dvc[i] = vc - v[i];
            v[i] = vc;
        }
        for (int k = 0; k < N; k++) {
            const double *A = qp->A + (size_t)k * 169, *B = qp->B + (size_t)k * 52;
            for (int i = 0; i < NX; i++) {
                double s = 0.0;
                for (int l = 0; l < NX; l++) s += A[i * NX + l] * dxs[(size_t)k * NX + l];
                for (int a = 0; a < NU; a++) s += B[i * NU + a] * dvc[k * 4 + a];
                dxs[(size_t)(k + 1) * NX + i] = s;
            }
        }
        double pi[NX], pn[NX];
        for (int i = 0; i < NX; i++) pi[i] = qp->QNd[i] * dxs[(size_t)N * NX + i];
        for (int k = N - 1; k >= 0; k--) {
            const double *A = qp->A + (size_t)k * 169, *B = qp->B + (size_t)k * 52;
            for (int a = 0; a < NU; a++) {
                double s = qp->Rd[a] * dvc[k * 4 + a];
                for (int l = 0; l < NX; l++) s += B[l * NU + a] * pi[l];
                gr[k * 4 + a] = s;
            }
            for (int i = 0; i < NX; i++) {
                double s = qp->Qd[i] * dxs[(size_t)k * NX + i];
                for (int l = 0; l < NX; l++) s += A[l * NX + i] * pi[l];
                pn[i] = s;
            }
            memcpy(pi, pn, sizeof pi);
        }
        double acc = 0.0;
        for (int i = 0; i < n; i++) {
            tl[i] = v[i] - qp->lb[i];
            tu[i] = qp->ub[i] - v[i];
            acc += fabs(gr[i]) * fmin(tl[i], tu[i]);
        }
        const double mu0 = fmax(o->lam0_min, o->mu0_scale * acc / n);
        for (int i = 0; i < n; i++) {
            ll[i] = fmax(gr[i], 0.0) + mu0 / tl[i];
            lu[i] = fmax(-gr[i], 0.0) + mu0 / tu[i];
            rg[i] = gr[i] - ll[i] + lu[i];
        }
        free(dxs);
    } else {
        const double mu0 = fmax(o->mu0_scale * viol, o->lam0_min);
        for (int i = 0; i < n; i++) {
            tl[i] = fmax(v[i] - qp->lb[i], o->thr0);
            tu[i] = fmax(qp->ub[i] - v[i], o->thr0);
            ll[i] = mu0 / tl[i];
            lu[i] = mu0 / tu[i];
            rg[i] = -ll[i] + lu[i];
        }
    }
    for (;;) {
        double mu = 0.0;
        res = 0.0;
        for (int i = 0; i < n; i++) {
            rl[i] = v[i] - qp->lb[i] - tl[i];
            ru[i] = qp->ub[i] - v[i] - tu[i];
            mu += ll[i] * tl[i] + lu[i] * tu[i];
            res = fmax(res, fmax(ll[i] * tl[i], lu[i] * tu[i]));
            res = fmax(res, fmax(fabs(rg[i]), fmax(fabs(rl[i]), fabs(ru[i]))));
        }
        mu /= nc;
        if (!(res == res)) { status = 4; break; }
        if (res <= o->tol) { status = 0; break; }
        if (it >= o->max_iter) break;
        it++;
        for (int i = 0; i < n; i++) {
            Dl[i] = ll[i] / tl[i];
            Du[i] = lu[i] / tu[i];
            Rhat[i] = qp->Rd[i % NU] + Dl[i] + Du[i];
            g[i] = rg[i] + ll[i] + Dl[i] * rl[i] - lu[i] - Du[i] * ru[i];
        }
        if (riccati_factor(qp, Rhat, g, 0)) { status = 4; break; }
        riccati_forward(qp, 0, dva, qp->x);
        for (int i = 0; i < n; i++) {
            dtl[i] = dva[i] + rl[i];
            dtu[i] = -dva[i] + ru[i];
            dll[i] = -ll[i] - Dl[i] * dtl[i];
            dlu[i] = -lu[i] - Du[i] * dtu[i];
        }
        double a = 1.0;
        a = steplen(n, tl, dtl, a); a = steplen(n, tu, dtu, a);
        a = steplen(n, ll, dll, a); a = steplen(n, lu, dlu, a);
        double mu_aff = 0.0;
        for (int i = 0; i < n; i++)
            mu_aff += (ll[i] + a * dll[i]) * (tl[i] + a * dtl[i]) + (lu[i] + a * dlu[i]) * (tu[i] + a * dtu[i]);
        mu_aff /= nc;
        const double sr = mu_aff / mu, smu = sr * sr * sr * mu;
        for (int i = 0; i < n; i++) {
            cl[i] = dll[i] * dtl[i];
            cu[i] = dlu[i] * dtu[i];
            g[i] = (cl[i] - smu) / tl[i] - (cu[i] - smu) / tu[i];
        }
        riccati_resolve(qp, g);
        riccati_forward(qp, 0, dvc, qp->x);
        for (int i = 0; i < n; i++) {
            const double dv = dva[i] + dvc[i];
            dva[i] = dv;
            dtl[i] = dv + rl[i];
            dtu[i] = -dv + ru[i];
            dll[i] = (smu - cl[i]) / tl[i] - ll[i] - Dl[i] * dtl[i];
            dlu[i] = (smu - cu[i]) / tu[i] - lu[i] - Du[i] * dtu[i];
        }
        a = 1.0;
        a = steplen(n, tl, dtl, a); a = steplen(n, tu, dtu, a);
        a = steplen(n, ll, dll, a); a = steplen(n, lu, dlu, a);
        a = fmin(1.0, o->tau * a);
        for (int i = 0; i < n; i++) {
            v[i] += a * dva[i];
            tl[i] += a * dtl[i]; tu[i] += a * dtu[i];
            ll[i] += a * dll[i]; lu[i] += a * dlu[i];
            rg[i] *= (1.0 - a);
        }
    }
    rollout(qp, v, qp->x);
    free(Rhat);
    *iters_out = it;
    *res_out = res;
    return status;
}

/* ---------------------------------------------------------------- public batch entry points */

/* Linearise one instance: fills A,B,b,q,r,lb,ub,dx0 of qp from iterate + references */
static void linearise(qp_t *qp, const cfo_opts *o, const double *xit, const double *uit,
                      const double *x0, const double *yref, const double *yref_e) {
    const int N = o->N;
    double phi[NX];
    for (int k = 0; k < N; k++) {
        const double *xk = xit + (size_t)k * NX, *uk = uit + (size_t)k * NU;
        cfo_rk4_sens(xk, uk, o->dt, phi, qp->A + (size_t)k * 169, qp->B + (size_t)k * 52);
        for (int i = 0; i < NX; i++) {
            qp->b[k * NX + i] = phi[i] - xit[(size_t)(k + 1) * NX + i];
            qp->q[k * NX + i] = o->W[i] * (xk[i] - yref[(size_t)k * NY + i]);
        }
        for (int i = 0; i < NU; i++) {
            qp->r[k * NU + i] = o->W[NX + i] * (uk[i] - yref[(size_t)k * NY + NX + i]);
            qp->lb[k * NU + i] = o->u_min - uk[i];
            qp->ub[k * NU + i] = o->u_max - uk[i];
        }
    }
    for (int i = 0; i < NX; i++) {
        qp->q[N * NX + i] = o->WN[i] * (xit[(size_t)N * NX + i] - yref_e[i]);
        qp->dx0[i] = x0[i] - xit[i];
    }
    qp->Qd = o->W; qp->Rd = o->W + NX; qp->QNd = o->WN;
}

/* One SQP-RTI step for B instances (acados_solve() equivalent, acados_mpc.cpp:611).
 *   x_it [B][N+1][13], u_it [B][N][4]  : persistent iterate, updated in place (full step)
 *   x0 [B][13], yref [B][N][17], yref_e [B][13]
 *   status [B] (0 ok, 2 maxiter, 4 QP failure), iters [B], res [B] (QP max-norm residual)
 *   nthreads <= 0 -> all OpenMP threads.  Returns number of threads used. */
int cfo_rti_step_w(const cfo_opts *o, int B, double *x_it, double *u_it, const double *x0,
                   const double *yref, const double *yref_e, int *status, int *iters, double *res,
                   int nthreads, unsigned char *wcls, int *wvalid);
int cfo_rti_step(const cfo_opts *o, int B, double *x_it, double *u_it, const double *x0,
                 const double *yref, const double *yref_e, int *status, int *iters, double *res,
                 int nthreads) {
    return cfo_rti_step_w(o, B, x_it, u_it, x0, yref, yref_e, status, iters, res, nthreads, NULL, NULL);
}
/* ... with the persistent warm-start state of the active set: wcls [B][N][4] bytes, wvalid [B] (both zero-initialised by
 * the caller before the first step; used when o->as_warm, may be NULL otherwise) */
int cfo_rti_step_w(const cfo_opts *o, int B, double *x_it, double *u_it, const double *x0,
                   const double *yref, const double *yref_e, int *status, int *iters, double *res,
                   int nthreads, unsigned char *wcls, int *wvalid) {
    const int N = o->N;
    int used = 1;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    used = nthreads;
#pragma omp parallel num_threads(nthreads)
#endif
    {
        double *mem = (double *)malloc(sizeof(double) * qp_doubles(N));
        qp_t qp;
        qp_carve(&qp, N, mem);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
        for (int n = 0; n < B; n++) {
            double *xi = x_it + (size_t)n * (N + 1) * NX, *ui = u_it + (size_t)n * N * NU;
            linearise(&qp, o, xi, ui, x0 + (size_t)n * NX, yref + (size_t)n * N * NY, yref_e + (size_t)n * NX);
            int it = 0;
            double rs = 0;
            const int use_w = o->as_warm && wcls && wvalid;
            qp.wcls = use_w ? wcls + (size_t)n * N * NU : NULL;
            qp.wvalid = use_w ? wvalid + n : NULL;
            const int st = ipm_solve(&qp, o, &it, &rs);
            for (int i = 0; i < (N + 1) * NX; i++) xi[i] += qp.x[i];
            for (int i = 0; i < N * NU; i++) ui[i] += qp.v[i];
            status[n] = st; iters[n] = it; res[n] = rs;
        }
        free(mem);
    }
    return used;
}

/* QP pieces for kernel-level parity tests: linearise one instance and return the blocks. */
void cfo_linearise(const cfo_opts *o, const double *x_it, const double *u_it, const double *x0,
                   const double *yref, const double *yref_e, double *A, double *Bm, double *b,
                   double *q, double *r) {
    const int N = o->N;
    double *mem = (double *)malloc(sizeof(double) * qp_doubles(N));
    qp_t qp;
    qp_carve(&qp, N, mem);
    linearise(&qp, o, x_it, u_it, x0, yref, yref_e);
    memcpy(A, qp.A, sizeof(double) * N * 169);
    memcpy(Bm, qp.B, sizeof(double) * N * 52);
    memcpy(b, qp.b, sizeof(double) * N * NX);
    memcpy(q, qp.q, sizeof(double) * (N + 1) * NX);
    memcpy(r, qp.r, sizeof(double) * N * NU);
    free(mem);
}

/* Solve the QP of one instance and return the step and multipliers (no iterate update). */
int cfo_qp_solve(const cfo_opts *o, const double *x_it, const double *u_it, const double *x0,
                 const double *yref, const double *yref_e, double *dx, double *du, double *lam_l,
                 double *lam_u, int *iters, double *res) {
    const int N = o->N;
    double *mem = (double *)malloc(sizeof(double) * qp_doubles(N));
    qp_t qp;
    qp_carve(&qp, N, mem);
    linearise(&qp, o, x_it, u_it, x0, yref, yref_e);
    memset(qp.ll, 0, sizeof(double) * N * NU);
    memset(qp.lu, 0, sizeof(double) * N * NU);
    const int st = ipm_solve(&qp, o, iters, res);
    memcpy(dx, qp.x, sizeof(double) * (N + 1) * NX);
    memcpy(du, qp.v, sizeof(double) * N * NU);
    memcpy(lam_l, qp.ll, sizeof(double) * N * NU);
    memcpy(lam_u, qp.lu, sizeof(double) * N * NU);
    free(mem);
    return st;
}

/* Batched predictor / plant step (acados_estimator.cpp:573-593): xn = RK4(x, u, T, steps) */
void cfo_sim(int B, const double *x, const double *u, double T, int steps, double *xn) {
#ifdef _OPENMP
#pragma omp parallel for
#endif
    for (int n = 0; n < B; n++) cfo_rk4(x + (size_t)n * NX, u + (size_t)n * NU, T, steps, xn + (size_t)n * NX);
}

/* Closed loop for the CPU baseline of bench.py: every instance runs `steps`
 * consecutive RTI steps against the model as plant (x <- RK4(x, u0, dt)), all inside ONE parallel
 * region -- one instance at a time per thread, static partition, so a thread's share is one
 * contiguous block of instances and the region is entered once (B-thr).  With B = 1 and
 * lat_us != NULL the wall time of every RTI step is recorded (B-lat: acados_solve() latency,
 * acados_mpc.cpp:611-616).  x [B][13] is the plant state (updated), iterate = hover start
 * (x_k = x, u_k = hover) as in the GPU benchmark.  Returns the threads used; *seconds = wall time
 * of the region; iters_sum = total QP solves / interior-point iterations; n_bad = steps whose
 * status was not 0. */
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
int cfo_closed_loop(const cfo_opts *o, int B, double *x, const double *yref, const double *yref_e, int steps,
                    int nthreads, double *seconds, long long *iters_sum, long long *n_bad, double *lat_us) {
    const int N = o->N;
    const double hov = sqrt((MQ * G0) / (4 * CT));
    int used = 1;
    long long it_tot = 0, bad_tot = 0;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#endif
    const double t_begin = now_s();
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads) reduction(+ : it_tot, bad_tot)
#endif
    {
#ifdef _OPENMP
#pragma omp single
        used = omp_get_num_threads();
#endif
        double *mem = (double *)malloc(sizeof(double) * (qp_doubles(N) + (size_t)(N + 1) * NX + (size_t)N * NU));
        qp_t qp;
        qp_carve(&qp, N, mem);
        double *xi = mem + qp_doubles(N), *ui = xi + (size_t)(N + 1) * NX;
        unsigned char *wc = (unsigned char *)calloc((size_t)N * NU, 1);
        int wv = 0;
        if (o->as_warm) { qp.wcls = wc; qp.wvalid = &wv; }
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int n = 0; n < B; n++) {
            double *xp = x + (size_t)n * NX;
            wv = 0;
            for (int k = 0; k <= N; k++) memcpy(xi + (size_t)k * NX, xp, sizeof(double) * NX);
            for (int i = 0; i < N * NU; i++) ui[i] = hov;
            for (int t = 0; t < steps; t++) {
                const double t0 = (lat_us && B == 1) ? now_s() : 0.0;
                linearise(&qp, o, xi, ui, xp, yref + (size_t)n * N * NY, yref_e + (size_t)n * NX);
                int it = 0;
                double rs = 0;
                const int st = ipm_solve(&qp, o, &it, &rs);
                for (int i = 0; i < (N + 1) * NX; i++) xi[i] += qp.x[i];
                for (int i = 0; i < N * NU; i++) ui[i] += qp.v[i];
                if (lat_us && B == 1) lat_us[t] = 1e6 * (now_s() - t0);
                it_tot += it;
                bad_tot += st != 0;
                double xn[NX];
                cfo_rk4(xp, ui, o->dt, 1, xn);
                memcpy(xp, xn, sizeof xn);
            }
        }
        free(wc);
        free(mem);
    }
    *seconds = now_s() - t_begin;
    if (iters_sum) *iters_sum = it_tot;
    if (n_bad) *n_bad = bad_tot;
    return used;
}
