// cfnmpc_ws.hpp -- device workspace description shared by the kernels and the C-ABI layer.
//
// HBM layout: every field is SoA with the instance index fastest,
//     field[(stage * E + elem) * Bp + inst],      Bp = batch rounded up to 64,
// FP64 throughout (the reference computes in double: BLASFEO d_ API, acados_mpc.cpp:68).
#pragma once
#include <hip/hip_runtime.h>

namespace cfn {

struct Params {
    int B, Bp, N;
    double dt;
    double W[17], WN[13];
    double u_min, u_max, tol, tau, thr0, lam0_min;
    int max_iter;
    // persistent iterate (acados nlp_out, acados_mpc.cpp:77) and per-step inputs
    double *xit;     // [(N+1)][13]
    double *uit;     // [N][4]
    double *x0;      // [1][13]
    double *yref;    // [N][17]
    double *yref_e;  // [1][13]
    // linearisation (written by k_linearise, read by every Riccati sweep)
    double *A;   // [N][97]  compact dPhi/dx (cfnmpc_model.hpp)
    double *Bm;  // [N][52]  dPhi/du, row-major 13x4
    double *b;   // [N][13]  Phi(x_k,u_k) - x_{k+1}
    // Riccati factors
    double *K;     // [N][52]  feedback gain, row-major 4x13
    double *Sinv;  // [N][10]  inverse of R^ + B'PB, packed symmetric
    double *d;     // [N][4]   feed-forward
    // interior-point state
    double *v, *tl, *tu, *ll, *lu, *rg, *dva, *dvc;  // [N][4] each
    int *status, *iters;
    double *res;
};

void launch_linearise(const Params& P, hipStream_t st);
void launch_qp(const Params& P, hipStream_t st);
void launch_sim(int B, const double* x, const double* u, double T, int steps, double* xn, hipStream_t st);
void launch_aos2soa(int B, int Bp, int S, int E, const double* aos, double* soa, hipStream_t st);
void launch_soa2aos(int B, int Bp, int S, int E, int s0, int Stot, const double* soa, double* aos, hipStream_t st);
void launch_init_iterate(const Params& P, int mode, hipStream_t st);

}  // namespace cfn
