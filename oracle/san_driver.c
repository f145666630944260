/* Sanitizer driver for the CPU restatement (TEST INFRASTRUCTURE ONLY): a few closed-loop RTI
 * steps of perturbed hover instances under -fsanitize=address,undefined.  Exit code 0 = clean. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int N; double dt; double W[17]; double WN[13]; double u_min, u_max; double tol; int max_iter;
    double tau, thr0, lam0_min, mu0_scale;
    int active_set;
    double clip_viol, clip_margin, as_skip_viol;
    int as_warm;
} cfo_opts;
void cfo_default_opts(cfo_opts *o);
int cfo_rti_step(const cfo_opts *o, int B, double *x_it, double *u_it, const double *x0, const double *yref,
                 const double *yref_e, int *status, int *iters, double *res, int nthreads);
int cfo_rti_step_w(const cfo_opts *o, int B, double *x_it, double *u_it, const double *x0, const double *yref,
                   const double *yref_e, int *status, int *iters, double *res, int nthreads, unsigned char *wcls, int *wvalid);
void cfo_sim(int B, const double *x, const double *u, double T, int steps, double *xn);

static int run(int active_set, int as_warm) {
    enum { B = 6, NX = 13, NU = 4, NY = 17 };
    cfo_opts o;
    cfo_default_opts(&o);
    o.active_set = active_set;
    o.as_warm = as_warm;
    const int N = o.N;
    unsigned char *wcls = calloc((size_t)B * N * NU, 1);
    int wvalid[B];
    memset(wvalid, 0, sizeof wvalid);
    double *xit = malloc(sizeof(double) * B * (N + 1) * NX), *uit = malloc(sizeof(double) * B * N * NU);
    double *yref = malloc(sizeof(double) * B * N * NY), yref_e[B * NX], x0[B * NX], xn[B * NX], u0[B * NU], res[B];
    int status[B], iters[B];
    const double hov = sqrt((33e-3 * 9.8066) / (4 * 3.25e-4));
    unsigned s = 12345u;
    for (int i = 0; i < B; i++) {
        double *x = x0 + i * NX;
        memset(x, 0, sizeof(double) * NX);
        x[3] = 1.0;
        for (int e = 0; e < 3; e++) { s = s * 1664525u + 1013904223u; x[e] = ((s >> 8) % 2000) / 1000.0 - 1.0; }
        x[2] += 0.4;
        for (int k = 0; k <= N; k++) memcpy(xit + ((size_t)i * (N + 1) + k) * NX, x, sizeof(double) * NX);
        for (int k = 0; k < N; k++) {
            for (int a = 0; a < NU; a++) uit[((size_t)i * N + k) * NU + a] = hov;
            double *y = yref + ((size_t)i * N + k) * NY;
            memset(y, 0, sizeof(double) * NY);
            y[2] = 0.4; y[3] = 1.0;
            for (int a = 0; a < NU; a++) y[NX + a] = hov;
        }
        memset(yref_e + i * NX, 0, sizeof(double) * NX);
        yref_e[i * NX + 2] = 0.4; yref_e[i * NX + 3] = 1.0;
    }
    int constrained = 0;
    for (int t = 0; t < 5; t++) {
        if (as_warm) cfo_rti_step_w(&o, B, xit, uit, x0, yref, yref_e, status, iters, res, 2, wcls, wvalid);
        else cfo_rti_step(&o, B, xit, uit, x0, yref, yref_e, status, iters, res, 2);
        for (int i = 0; i < B; i++) {
            if (status[i] != 0) { fprintf(stderr, "status %d\n", status[i]); return 2; }
            constrained += iters[i] > 0;
            memcpy(u0 + i * NU, uit + (size_t)i * N * NU, sizeof(double) * NU);
        }
        cfo_sim(B, x0, u0, 0.015, 1, xn);
        memcpy(x0, xn, sizeof xn);
    }
    printf("sanitizer run ok (active_set = %d, as_warm = %d), %d constrained solves\n", active_set, as_warm, constrained);
    free(xit); free(uit); free(yref); free(wcls);
    return constrained > 0 ? 0 : 3;
}

int main(void) {
    const int a = run(0, 0), b = run(1, 0), c = run(1, 1);   /* interior point, active-set solves, ... warm-started */
    return a ? a : (b ? b : c);
}
