/* batch_hover.c -- the batch C-ABI (include/cfnmpc.h) from plain C99, no Python and no HIP headers:
 * a fleet of B vehicles regulated to a hover point in closed loop through the library's RK4 plant,
 * the way NMPC::iteration drives ONE vehicle through acados_solve (acados_mpc.cpp:427-670).
 *
 *   gcc -std=c99 -O2 -Iinclude examples/batch_hover.c -Lcrazyflie_nmpc_amd -lcfnmpc \
 *       -Wl,-rpath,$PWD/crazyflie_nmpc_amd -lm -o batch_hover && ./batch_hover 1024 40
 *
 * Prints one line per 10 steps (worst position error of the fleet, share of saturated QPs) and a final
 * `OK ...` line; exit status 0 iff every solve of the last step returned status 0 and -- for runs of 300
 * steps (4.5 s) or more -- every vehicle ended within 2 cm of the target. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "cfnmpc.h"

#define CHECK(call)                                                          \
    do {                                                                     \
        int rc_ = (call);                                                    \
        if (rc_ != CFNMPC_OK) {                                              \
            fprintf(stderr, "%s failed with %d (%s)\n", #call, rc_, cfnmpc_version()); \
            return 2;                                                        \
        }                                                                    \
    } while (0)

static double unit(unsigned *s) { /* small LCG: the example must not depend on anything but libc */
    *s = *s * 1664525u + 1013904223u;
    return (double)(*s >> 8) / 16777216.0;
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 1024;
    const int steps = argc > 2 ? atoi(argv[2]) : 40;
    cfnmpc_opts o;
    if (CFNMPC_DEFAULT_OPTS(&o) != CFNMPC_OK) { /* the generator's constants: N = 50, dt = 15 ms, W, box 0..22 kRPM */
        fprintf(stderr, "libcfnmpc.so was built for another cfnmpc_opts (ABI %d, this header: %d)\n", cfnmpc_abi_version(), CFNMPC_ABI_VERSION);
        return 1;
    }
    const int N = o.N;
    const double hov = 15.777730167256925; /* sqrt(mq g0 / (4 Ct)), generate_c_code.py:59 */
    const double target[3] = {0.0, 0.0, 0.4};

    double *x = malloc(sizeof(double) * B * 13), *xn = malloc(sizeof(double) * B * 13);
    double *u0 = malloc(sizeof(double) * B * 4);
    double *yref = malloc(sizeof(double) * (size_t)B * N * 17), *yref_e = malloc(sizeof(double) * B * 13);
    int *status = malloc(sizeof(int) * B), *iters = malloc(sizeof(int) * B);
    if (!x || !xn || !u0 || !yref || !yref_e || !status || !iters) return 2;

    unsigned seed = 20200101u;
    for (int i = 0; i < B; i++) {
        double *xi = x + 13 * i;
        for (int j = 0; j < 13; j++) xi[j] = 0.0;
        for (int j = 0; j < 3; j++) xi[j] = target[j] + 0.6 * (unit(&seed) - 0.5); /* +-0.3 m */
        xi[3] = 1.0;                                                             /* level attitude */
        for (int j = 7; j < 10; j++) xi[j] = unit(&seed) - 0.5;                    /* +-0.5 m/s */
        for (int k = 0; k < N; k++) {                                             /* regulation rows, acados_mpc.cpp:435-454 */
            double *r = yref + ((size_t)i * N + k) * 17;
            for (int j = 0; j < 17; j++) r[j] = 0.0;
            r[0] = target[0]; r[1] = target[1]; r[2] = target[2]; r[3] = 1.0;
            r[13] = r[14] = r[15] = r[16] = hov;
        }
        for (int j = 0; j < 13; j++) yref_e[13 * i + j] = yref[(size_t)i * N * 17 + j];
    }

    cfnmpc_solver *s = NULL;
    CHECK(cfnmpc_create(&s, B, &o));
    CHECK(cfnmpc_set_yref(s, yref, yref_e, 0, NULL));
    CHECK(cfnmpc_set_x0(s, x, 0, NULL));
    CHECK(cfnmpc_init_iterate(s, CFNMPC_INIT_HOVER, NULL));
    printf("%s: %d vehicles, N = %d, %.1f MB of device workspace\n", cfnmpc_version(), cfnmpc_batch(s), cfnmpc_horizon(s),
           (double)cfnmpc_workspace_bytes(s) / 1e6);

    int bad = 0;
    double worst = 0.0;
    for (int t = 0; t < steps; t++) {
        CHECK(cfnmpc_set_x0(s, x, 0, NULL));               /* "lbx"/"ubx" of stage 0 */
        CHECK(cfnmpc_solve(s, 1, NULL));                   /* acados_solve() for every vehicle */
        CHECK(cfnmpc_get_u(s, 0, u0, 0, NULL));            /* "u" of stage 0 */
        CHECK(cfnmpc_get_stats(s, status, iters, NULL, 0, NULL));
        CHECK(cfnmpc_sim(B, x, u0, o.dt, 1, xn, 0, NULL)); /* the plant: one RK4 step */
        double *tmp = x; x = xn; xn = tmp;
        int saturated = 0;
        bad = 0; worst = 0.0;
        for (int i = 0; i < B; i++) {
            bad += status[i] != 0;
            saturated += iters[i] > 0;
            double e = 0.0;
            for (int j = 0; j < 3; j++) e = fmax(e, fabs(x[13 * i + j] - target[j]));
            worst = fmax(worst, e);
        }
        if (t % 10 == 9 || t == steps - 1)
            printf("step %3d: worst |p - target| = %.4f m, QPs with active bounds %5.1f %%, status != 0: %d\n", t + 1, worst,
                   100.0 * saturated / B, bad);
    }
    CHECK(cfnmpc_free(s));
    free(x); free(xn); free(u0); free(yref); free(yref_e); free(status); free(iters);
    const int converged = worst < 0.02;
    if (bad == 0 && (steps < 300 || converged)) printf("OK %s after %d steps\n", converged ? "converged" : "running", steps);
    return (bad == 0 && (steps < 300 || converged)) ? 0 : 1;
}
