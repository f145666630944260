// cf_nmpc_replay.cpp -- TEST HARNESS: ROS-free closed-loop replay of the reference node's per-step protocol
// through the acados-named drop-in (libacados_solver_crazyflie.so).  It plays the part of
// crazyflie_controller/src/acados_mpc.cpp's main(): it DEFINES the acados globals the
// generated solver expects from its caller (acados_mpc.cpp:76-84) and drives NMPC::iteration().
// The plant is the model itself, integrated by the sim solver (crazyflie_acados_sim_solve).
//
// usage: cf_nmpc_replay <regulation|tracking> <traj.txt|-> <steps> <x0.txt> <init 0|1> <out.csv> [uss [fixed_u0]]
//   uss: steady-state propeller speed of the hold rows (default: the node's own float value)
//   fixed_u0 = 1: the reference's FIXED_U0 variant (acados_mpc.cpp:605-608, 631-635): stage 0 pinned to the input in
//   flight (lbu = ubu = u1 of the previous step), u1 is what goes to the motors
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <fstream>
#include <vector>

#include "../../include/acados_sim_solver_crazyflie.h"
#include "cf_nmpc_node.hpp"

// global data, exactly as the reference node declares it (acados_mpc.cpp:76-84)
ocp_nlp_in* nlp_in;
ocp_nlp_out* nlp_out;
ocp_nlp_solver* nlp_solver;
void* nlp_opts;
ocp_nlp_plan* nlp_solver_plan;
ocp_nlp_config* nlp_config;
ocp_nlp_dims* nlp_dims;
external_function_param_casadi* forw_vde_casadi;

int main(int argc, char** argv) {
    if (argc < 7) {
        std::fprintf(stderr, "usage: %s <regulation|tracking> <traj.txt|-> <steps> <x0.txt> <init 0|1> <out.csv>\n", argv[0]);
        return 2;
    }
    const bool tracking = !std::strcmp(argv[1], "tracking");
    const std::string traj = std::strcmp(argv[2], "-") ? argv[2] : "";
    const int steps = std::atoi(argv[3]);
    double x[13];
    {
        std::ifstream f(argv[4]);
        for (double& v : x) f >> v;
        if (!f) { std::fprintf(stderr, "cannot read 13 numbers from %s\n", argv[4]); return 2; }
    }
    const int init = std::atoi(argv[5]);
    std::FILE* out = std::fopen(argv[6], "w");
    if (!out) return 2;

    cf::NMPC nmpc(traj);
    if (argc > 7) nmpc.uss_row = std::atof(argv[7]);
    if (argc > 8) { nmpc.fixed_u0 = std::atoi(argv[8]) != 0; for (double& v : nmpc.acados_out.u1) v = nmpc.uss_row; }
    if (nlp_out == nullptr || nlp_dims == nullptr || nlp_dims->N != cf::N) {
        std::fprintf(stderr, "acados_create() did not populate the caller's globals\n");
        return 3;
    }
    if (tracking) nmpc.reconfigure(true, false, 0, 0, 0);
    else nmpc.reconfigure(false, true, 0.0, 0.0, 0.40);
    if (crazyflie_acados_sim_create()) return 3;
    if (init == 1) {
        ocp_nlp_constraints_model_set(nlp_config, nlp_dims, nlp_in, 0, "lbx", x);
        ocp_nlp_constraints_model_set(nlp_config, nlp_dims, nlp_in, 0, "ubx", x);
        if (acados_cfnmpc_init_iterate(1)) return 3;
    }
    double Ts = 0.015;
    std::vector<double> solve_ms;   // wall time of acados_solve() per step (nlp_out->total_time, acados_mpc.cpp:616)
    for (int t = 0; t < steps; t++) {
        cf::CrazyflieState msg;
        for (int i = 0; i < 3; i++) { msg.pos[i] = x[i]; msg.vel[i] = x[7 + i]; msg.rates[i] = x[10 + i]; }
        for (int i = 0; i < 4; i++) msg.quat[i] = x[3 + i];
        const int status = nmpc.iteration(msg);
        std::fprintf(out, "%d,%d,%d", t, status, (int)nmpc.policy);
        for (double v : nmpc.acados_out.u0) std::fprintf(out, ",%.17g", v);
        for (double v : nmpc.acados_out.u1) std::fprintf(out, ",%.17g", v);
        for (double v : nmpc.acados_out.x4) std::fprintf(out, ",%.17g", v);
        std::fprintf(out, ",%.17g,%.17g,%.17g,%.17g", nmpc.last_cmd_vel.linear_x, nmpc.last_cmd_vel.linear_y,
                     nmpc.last_cmd_vel.linear_z, nmpc.last_cmd_vel.angular_z);
        std::fprintf(out, ",%d,%d,%d,%d", nmpc.last_motvel.w1, nmpc.last_motvel.w2, nmpc.last_motvel.w3, nmpc.last_motvel.w4);
        std::fprintf(out, ",%.6g,%d\n", nmpc.acados_out.KKT_res, nlp_out->qp_iter);
        solve_ms.push_back(1e3 * nmpc.acados_out.cpu_time);
        // plant: one sampling period of the model with u0 (FP64 u, App. B1)
        sim_in_set(crazyflie_sim_config, crazyflie_sim_dims, crazyflie_sim_in, "T", &Ts);
        sim_in_set(crazyflie_sim_config, crazyflie_sim_dims, crazyflie_sim_in, "x", x);
        sim_in_set(crazyflie_sim_config, crazyflie_sim_dims, crazyflie_sim_in, "u", nmpc.acados_out.u0);
        crazyflie_sim_config->num_steps = 1;
        if (crazyflie_acados_sim_solve()) return 4;
        sim_out_get(crazyflie_sim_config, crazyflie_sim_dims, crazyflie_sim_out, "xn", x);
    }
    std::fclose(out);
    if (!solve_ms.empty()) {   // config C1: single-instance latency against the node's 15 ms period
        std::sort(solve_ms.begin(), solve_ms.end());
        std::fprintf(stderr, "acados_solve() wall time over %zu steps [ms]: median %.3f  p90 %.3f  max %.3f\n", solve_ms.size(),
                     solve_ms[solve_ms.size() / 2], solve_ms[solve_ms.size() * 9 / 10], solve_ms.back());
    }
    crazyflie_acados_sim_free();
    return 0;
}
