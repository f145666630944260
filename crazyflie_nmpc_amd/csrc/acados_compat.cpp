// acados_compat.cpp -- the acados-named drop-in boundary (include/acados_solver_crazyflie.h,
// include/acados_sim_solver_crazyflie.h) over the batch engine with batch = 1.
//
// Call protocol it serves (crazyflie_controller/src/acados_mpc.cpp):
//   acados_create()                              :225
//   ocp_nlp_constraints_model_set(.. "lbx"/"ubx")   :581-582
//   ocp_nlp_cost_model_set(.. k, "yref")            :590-594      ("W" :599-601 if SET_WEIGHTS)
//   acados_solve()                                :611
//   nlp_out->inf_norm_res / total_time            :615-616
//   ocp_nlp_out_get(.. "u"/"x")                    :619-625
// and the estimator's predictor (acados_estimator.cpp:237, 573-593).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "../../include/acados_sim_solver_crazyflie.h"
#include "../../include/acados_solver_crazyflie.h"
#include "../../include/cfnmpc.h"

namespace {
constexpr int N = CRAZYFLIE_N, NX = CRAZYFLIE_NX, NU = CRAZYFLIE_NU, NY = CRAZYFLIE_NY;

struct Shim {
    cfnmpc_solver* s = nullptr;
    cfnmpc_opts opts;
    double lbx[NX], ubx[NX];
    double yref[N * NY], yref_e[NX];
    double W[NY], WN[NX];
    double x[(N + 1) * NX], u[N * NU];  // host copy of the iterate after the last solve
    double lbu[N][NU], ubu[N][NU];      // per-stage input box as set through "lbu" / "ubu"
    bool weights_dirty = false, box_dirty = false;
    ocp_nlp_in in;
    ocp_nlp_out out;
    ocp_nlp_solver solver;
    ocp_nlp_plan plan;
    ocp_nlp_config config;
    ocp_nlp_dims dims;
    external_function_param_casadi vde;
};
Shim* g = nullptr;

sim_config g_sim_config;
sim_in g_sim_in;
sim_out g_sim_out;
bool g_sim_ready = false;
}  // namespace

extern "C" {

// Weak definitions: the reference node defines these globals itself (acados_mpc.cpp:76-84);
// its strong definitions win at link time.  Stand-alone users (tests, ctypes) get these.
__attribute__((weak)) ocp_nlp_in* nlp_in = nullptr;
__attribute__((weak)) ocp_nlp_out* nlp_out = nullptr;
__attribute__((weak)) ocp_nlp_solver* nlp_solver = nullptr;
__attribute__((weak)) void* nlp_opts = nullptr;
__attribute__((weak)) ocp_nlp_plan* nlp_solver_plan = nullptr;
__attribute__((weak)) ocp_nlp_config* nlp_config = nullptr;
__attribute__((weak)) ocp_nlp_dims* nlp_dims = nullptr;
__attribute__((weak)) external_function_param_casadi* forw_vde_casadi = nullptr;

sim_config* crazyflie_sim_config = nullptr;
void* crazyflie_sim_dims = nullptr;
sim_in* crazyflie_sim_in = nullptr;
sim_out* crazyflie_sim_out = nullptr;

int acados_create(void) {
    if (g) return 0;
    Shim* h = new Shim();
    if (CFNMPC_DEFAULT_OPTS(&h->opts) != CFNMPC_OK) { delete h; return 1; }   // shim and engine built from different headers
    if (cfnmpc_create(&h->s, 1, &h->opts) != CFNMPC_OK) {
        delete h;
        return 1;
    }
    // codegen defaults: x0 = [0,0,0,1,0..] (generate_c_code.py:135), yref = [0,0,0.5,1,0..,hov_w x4],
    // yref_e = [0,0,0.5,1,0..] (generate_c_code.py:128-129)
    const double hov = std::sqrt((33e-3 * 9.8066) / (4 * 3.25e-4));
    std::memset(h->lbx, 0, sizeof h->lbx);
    h->lbx[3] = 1.0;
    std::memcpy(h->ubx, h->lbx, sizeof h->lbx);
    for (int k = 0; k < N; k++) {
        double* r = h->yref + k * NY;
        std::memset(r, 0, sizeof(double) * NY);
        r[2] = 0.5; r[3] = 1.0;
        for (int i = 0; i < NU; i++) r[NX + i] = hov;
    }
    std::memset(h->yref_e, 0, sizeof h->yref_e);
    h->yref_e[2] = 0.5; h->yref_e[3] = 1.0;
    for (int i = 0; i < NY; i++) h->W[i] = h->opts.W[i];
    for (int i = 0; i < NX; i++) h->WN[i] = h->opts.WN[i];
    for (int k = 0; k < N; k++)
        for (int i = 0; i < NU; i++) { h->lbu[k][i] = h->opts.u_min; h->ubu[k][i] = h->opts.u_max; }
    for (int k = 0; k <= N; k++) { std::memset(h->x + k * NX, 0, sizeof(double) * NX); h->x[k * NX + 3] = 1.0; }
    std::memset(h->u, 0, sizeof h->u);
    h->dims = ocp_nlp_dims{N, NX, NU, NY, NX};
    h->config.N = N;
    h->plan.nlp_solver = 1;  // SQP_RTI
    h->in.priv = h; h->solver.priv = h; h->vde.priv = nullptr;
    h->out = ocp_nlp_out{0.0, 0.0, 1, 0, h};
    g = h;
    nlp_in = &h->in; nlp_out = &h->out; nlp_solver = &h->solver; nlp_opts = &h->opts;
    nlp_solver_plan = &h->plan; nlp_config = &h->config; nlp_dims = &h->dims; forw_vde_casadi = &h->vde;
    return 0;
}

int acados_free(void) {
    if (!g) return 0;
    cfnmpc_free(g->s);
    delete g;
    g = nullptr;
    nlp_in = nullptr; nlp_out = nullptr; nlp_solver = nullptr; nlp_opts = nullptr;
    nlp_solver_plan = nullptr; nlp_config = nullptr; nlp_dims = nullptr; forw_vde_casadi = nullptr;
    return 0;
}

int acados_cfnmpc_init_iterate(int mode) {
    if (!g) return 1;
    if (cfnmpc_set_x0(g->s, g->lbx, 0, nullptr) != CFNMPC_OK) return 1;
    if (cfnmpc_init_iterate(g->s, mode, nullptr) != CFNMPC_OK) return 1;
    // ocp_nlp_out_get reads the host copy of the iterate: refresh it
    return cfnmpc_get_iterate(g->s, g->x, g->u, 0, nullptr) == CFNMPC_OK ? 0 : 1;
}

int ocp_nlp_constraints_model_set(ocp_nlp_config*, ocp_nlp_dims*, ocp_nlp_in*, int stage, const char* field,
                                  void* value) {
    if (!g || !field || !value) return 1;
    const double* v = static_cast<const double*>(value);
    if (!std::strcmp(field, "lbx") || !std::strcmp(field, "ubx")) {
        if (stage != 0) return 1;  // only the initial state is constrained (generate_c_code.py:131-136)
        std::memcpy(field[0] == 'l' ? g->lbx : g->ubx, v, sizeof(double) * NX);
        return 0;
    }
    if (!std::strcmp(field, "lbu") || !std::strcmp(field, "ubu")) {
        // stored per stage like acados does; acados_solve() hands them to the engine: as ONE scalar box when
        // they are uniform (the default path), as per-stage / per-input boxes otherwise -- e.g. the
        // reference's FIXED_U0 pin of stage 0, lbu = ubu = u1 (acados_mpc.cpp:605-608, compiled out at :111)
        if (stage < 0 || stage >= N) return 1;
        for (int i = 0; i < NU; i++) if (!(v[i] == v[i])) return 1;
        std::memcpy(field[0] == 'l' ? g->lbu[stage] : g->ubu[stage], v, sizeof(double) * NU);
        g->box_dirty = true;
        return 0;
    }
    return 1;
}

int ocp_nlp_cost_model_set(ocp_nlp_config*, ocp_nlp_dims*, ocp_nlp_in*, int stage, const char* field, void* value) {
    if (!g || !field || !value || stage < 0 || stage > N) return 1;
    const double* v = static_cast<const double*>(value);
    if (!std::strcmp(field, "yref")) {
        if (stage < N) std::memcpy(g->yref + stage * NY, v, sizeof(double) * NY);
        else std::memcpy(g->yref_e, v, sizeof(double) * NX);
        return 0;
    }
    if (!std::strcmp(field, "W")) {
        // the node writes only the diagonal (acados_mpc.cpp:526-556): element i + i*n is the same
        // in row- and column-major order
        // (state weights may be 0 -- config/crazyflie_params.cfg ranges start at 0.0 --, input
        //  weights must be positive; the whole diagonal is checked before anything is copied)
        const int n = stage < N ? NY : NX;
        double* dst = stage < N ? g->W : g->WN;
        for (int i = 0; i < n; i++) {
            const double w = v[i + i * n];
            if (!(i < NX ? w >= 0.0 : w > 0.0)) return 1;
        }
        for (int i = 0; i < n; i++) dst[i] = v[i + i * n];
        g->weights_dirty = true;
        return 0;
    }
    return 1;
}

int acados_solve(void) {
    if (!g) return 1;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < NX; i++)
        if (g->lbx[i] != g->ubx[i]) return 1;  // x0 must be pinned: lbx == ubx (acados_mpc.cpp:581-582)
    if (g->weights_dirty) {
        if (cfnmpc_set_weights(g->s, g->W, g->WN) != CFNMPC_OK) return 1;
        g->weights_dirty = false;
    }
    if (g->box_dirty) {
        const double lo = g->lbu[0][0], hi = g->ubu[0][0];
        bool uniform = true;
        for (int k = 0; k < N; k++)
            for (int i = 0; i < NU; i++)
                if (g->lbu[k][i] != lo || g->ubu[k][i] != hi) uniform = false;
        if (uniform) {
            if (cfnmpc_set_box(g->s, lo, hi) != CFNMPC_OK) return 1;          // (also rejects lo >= hi)
            if (cfnmpc_set_box_stages(g->s, nullptr, nullptr, 0, nullptr) != CFNMPC_OK) return 1;
        } else {
            // [1][N][4] = the shim's own storage order (rejects lb > ub and NaN; lb = ub pins an input)
            if (cfnmpc_set_box_stages(g->s, &g->lbu[0][0], &g->ubu[0][0], 0, nullptr) != CFNMPC_OK) return 1;
        }
        g->box_dirty = false;
    }
    int status = 1, iters = 0;
    double res = 0.0;
    // inputs in, one RTI step, iterate and statistics out: one transfer each way, one synchronisation
    if (cfnmpc_step_host(g->s, g->lbx, g->yref, g->yref_e, g->u, g->x, &status, &iters, &res, nullptr) != CFNMPC_OK) return 1;
    g->out.inf_norm_res = res;
    g->out.qp_iter = iters;
    g->out.total_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return status;
}

void ocp_nlp_out_get(ocp_nlp_config*, ocp_nlp_dims*, ocp_nlp_out*, int stage, const char* field, void* value) {
    if (!g || !field || !value) return;
    double* v = static_cast<double*>(value);
    if (!std::strcmp(field, "x") && stage >= 0 && stage <= N) std::memcpy(v, g->x + stage * NX, sizeof(double) * NX);
    else if (!std::strcmp(field, "u") && stage >= 0 && stage < N) std::memcpy(v, g->u + stage * NU, sizeof(double) * NU);
}

// ---------------------------------------------------------------- predictor (sim solver)
int crazyflie_acados_sim_create(void) {
    std::memset(&g_sim_in, 0, sizeof g_sim_in);
    std::memset(&g_sim_out, 0, sizeof g_sim_out);
    g_sim_in.T = 0.06;  // launch/acados_predictor.launch:62
    g_sim_in.x[3] = 1.0;
    g_sim_config.ns = 4;         // stages of the explicit RK scheme (classic RK4)
    g_sim_config.num_steps = 4;  // integration steps over T (4 x 15 ms for T = 60 ms)
    crazyflie_sim_config = &g_sim_config;
    crazyflie_sim_dims = &g_sim_config;
    crazyflie_sim_in = &g_sim_in;
    crazyflie_sim_out = &g_sim_out;
    g_sim_ready = true;
    return 0;
}

int crazyflie_acados_sim_free(void) {
    g_sim_ready = false;
    crazyflie_sim_config = nullptr; crazyflie_sim_dims = nullptr; crazyflie_sim_in = nullptr; crazyflie_sim_out = nullptr;
    return 0;
}

int sim_in_set(void*, void*, sim_in* in, const char* field, void* value) {
    if (!in || !field || !value) return 1;
    const double* v = static_cast<const double*>(value);
    if (!std::strcmp(field, "T")) { in->T = v[0]; return 0; }
    if (!std::strcmp(field, "x")) { std::memcpy(in->x, v, sizeof in->x); return 0; }
    if (!std::strcmp(field, "u")) { std::memcpy(in->u, v, sizeof in->u); return 0; }
    return 1;
}

int sim_out_get(void*, void*, sim_out* out, const char* field, void* value) {
    if (!out || !field || !value) return 1;
    if (!std::strcmp(field, "xn") || !std::strcmp(field, "x")) { std::memcpy(value, out->xn, sizeof out->xn); return 0; }
    return 1;
}

int crazyflie_acados_sim_solve(void) {
    if (!g_sim_ready) return 1;
    const auto t0 = std::chrono::steady_clock::now();
    if (!(g_sim_in.T > 0.0)) return 1;
    if (g_sim_config.ns != 4 || g_sim_config.num_steps < 1) return 1;   // only the 4-stage scheme exists
    const int rc = cfnmpc_sim(1, g_sim_in.x, g_sim_in.u, g_sim_in.T, g_sim_config.num_steps, g_sim_out.xn, 0, nullptr);
    g_sim_out.total_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return rc == CFNMPC_OK ? 0 : 1;
}

}  // extern "C"
