// cf_nmpc_node.hpp -- TEST HARNESS (not part of the product package): ROS-free host-side mirror of the reference NMPC node
// (crazyflie_controller/src/acados_mpc.cpp, class NMPC :115-719) on top of the acados-named
// C-ABI of include/acados_solver_crazyflie.h.  Same member / method names and per-step protocol
// as the reference; ROS messages are replaced by plain structs with the message layouts
// (msg/CrazyflieStateStamped.msg, msg/PropellerSpeedsStamped.msg, geometry_msgs/Twist).
//
// Deliberate differences (all flagged where they occur):
//   B4  `policy` is initialised (Regulation, z = 0.40 as the dynamic_reconfigure default,
//       config/crazyflie_params.cfg:12,17) instead of relying on the server's first callback;
//   B6  the solver status is kept in `acados_status` AND returned by iteration();
//   Eigen's Quaterniond is replaced by four doubles (only normalize() was used, :650).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/acados_solver_crazyflie.h"   // the drop-in header under test

namespace cf {

// acados dims (acados_mpc.cpp:96-104)
constexpr int N = 50, NX = 13, NU = 4, NY = 17, NYN = 13;
constexpr double pi = 3.14159265358979323846;
constexpr double g0 = 9.80665;  // the node's constant (acados_mpc.cpp:107); the model uses 9.8066 (App. B2)

struct CrazyflieState {  // msg/CrazyflieStateStamped.msg
    double pos[3], vel[3], quat[4] /* w x y z */, rates[3];
};
struct PropellerSpeeds {  // msg/PropellerSpeedsStamped.msg: int32 w1..w4 (truncation, App. B1)
    int32_t w1, w2, w3, w4;
};
struct Twist {  // geometry_msgs/Twist fields the node fills (acados_mpc.cpp:655-668)
    double linear_x /* pitch deg */, linear_y /* -roll deg */, linear_z /* thrust PWM */, angular_z /* yaw rate deg/s */;
};

class NMPC {
public:
    enum systemStates { xq = 0, yq, zq, qw, qx, qy, qz, vbx, vby, vbz, wx, wy, wz };
    enum controlInputs { w1 = 0, w2, w3, w4 };
    enum reference_mode { Regulation = 0, Tracking = 1, Position_Hold = 2 };
    struct euler { double phi, theta, psi; };
    struct solver_output {
        double status, KKT_res, cpu_time;
        double u0[NU], u1[NU], x4[NX];
    };
    struct solver_input {
        double x0[NX], yref[NY * N], yref_e[NYN];
    };

    float uss, Ct, mq;  // float as in the reference (acados_mpc.cpp:189)
    double uss_row;     // value written into the hold rows: uss unless the harness overrides it (App. B2)
    double x0_sign[NX], yref_sign[(NY * N) + NY];
    double xq_des, yq_des, zq_des;
    solver_input acados_in;
    solver_output acados_out;
    int acados_status;
    reference_mode policy;
    std::vector<double> precomputed_traj;   // [N_STEPS][NY], row-major (the node keeps a vector of rows)
    int N_STEPS, iter;
    PropellerSpeeds last_motvel;  // what would go to /crazyflie/acados_motvel
    Twist last_cmd_vel;           // what would go to /crazyflie/cmd_vel
    bool fixed_u0;                // the reference's compile-time switch FIXED_U0 (acados_mpc.cpp:111), run-time here

    // acados_mpc.cpp:219-291
    explicit NMPC(const std::string& ref_traj) {
        const int status = acados_create();
        if (status) throw status;  // the reference exit(1)s (:227-230)
        for (int i = 0; i < NU; i++) acados_out.u0[i] = 0.0;
        fixed_u0 = false;
        mq = 33e-3f;
        Ct = 3.25e-4f;
        uss = std::sqrt((mq * g0) / (4 * Ct));
        uss_row = uss;
        for (int i = 0; i < NU; i++) acados_out.u1[i] = uss;   // (the reference leaves u1 uninitialised before the first solve)
        N_STEPS = ref_traj.empty() ? 0 : load_reference(ref_traj.c_str(), precomputed_traj);
        xq_des = 0; yq_des = 0; zq_des = 0.40;  // App. B4
        iter = 0;
        policy = Regulation;
        acados_status = 0;
    }
    ~NMPC() { nmpcReset(); }

    // acados_mpc.cpp:305-329 (dynamic_reconfigure callback), reduced to its effect
    void reconfigure(bool enable_traj_tracking, bool enable_regulation, double x, double y, double z) {
        if (enable_traj_tracking) policy = Tracking;
        if (enable_regulation) { xq_des = x; yq_des = y; zq_des = z; policy = Regulation; }
    }

    // Trajectory file of the Tracking policy: one reference row of NY whitespace-separated numbers
    // per line, one line per sampling period; the row count is the line count (loader semantics of
    // acados_mpc.cpp:354-382).  Rows land back to back in `rows`; short lines are zero-padded.
    static int load_reference(const char* path, std::vector<double>& rows) {
        std::FILE* f = std::fopen(path, "r");
        if (!f) return 0;
        int n_lines = 0;
        char buf[4096];
        while (std::fgets(buf, sizeof buf, f)) {
            rows.resize((size_t)(n_lines + 1) * NY, 0.0);
            char* cur = buf;
            for (int j = 0; j < NY; j++) {
                char* end = nullptr;
                const double v = std::strtod(cur, &end);
                if (end == cur) break;
                rows[(size_t)n_lines * NY + j] = v;
                cur = end;
            }
            n_lines++;
        }
        std::fclose(f);
        return n_lines;
    }

    // Roll / pitch / heading of a unit quaternion (w x y z) in the node's convention
    // (acados_mpc.cpp:384-404): roll and pitch from the third row of the rotation matrix, heading
    // from its first column.
    static euler attitude(const double (&q)[4]) {
        const double ww = q[0] * q[0];
        euler e;
        e.phi = std::atan2(2 * (q[2] * q[3] - q[0] * q[1]), 2 * (ww + q[3] * q[3]) - 1);
        e.theta = -std::asin(2 * (q[1] * q[3] + q[0] * q[2]));
        e.psi = std::atan2(2 * (q[1] * q[2] - q[0] * q[3]), 2 * (ww + q[1] * q[1]) - 1);
        return e;
    }
    static double rad2Deg(double rad) { return rad * 180.0 / pi; }
    // acados_mpc.cpp:421-425 (truncation to int is the wire unit, App. B8)
    static int krpm2pwm(double Krpm) { return (int)(((Krpm * 1000) - 4070.3) / 0.2685); }

    void nmpcReset() { acados_free(); }  // acados_mpc.cpp:416-419

    // acados_mpc.cpp:427-718; returns the solver status (App. B6)
    int iteration(const CrazyflieState& msg) {
        // without a usable trajectory the reference would index an empty vector (:460-513): hold instead
        if (policy != Regulation && N_STEPS < N + 1) policy = Regulation;
        switch (policy) {
            case Regulation:
                for (int k = 0; k < N + 1; k++) fill_hold_row(k, xq_des, yq_des, zq_des);
                break;
            case Tracking:
                if (iter < N_STEPS - N) {
                    for (int k = 0; k < N + 1; k++)
                        for (int j = 0; j < NY; j++) yref_sign[k * NY + j] = row(iter + k)[j];
                    ++iter;
                } else {
                    policy = Position_Hold;  // the reference keeps the previous window for this step (:486)
                }
                break;
            case Position_Hold:
                for (int k = 0; k < N + 1; k++)
                    fill_hold_row(k, row(N_STEPS - 1)[xq], row(N_STEPS - 1)[yq], row(N_STEPS - 1)[zq]);
                break;
        }
        // --- read estimate (:560-578)
        acados_in.x0[xq] = msg.pos[0]; acados_in.x0[yq] = msg.pos[1]; acados_in.x0[zq] = msg.pos[2];
        acados_in.x0[qw] = msg.quat[0]; acados_in.x0[qx] = msg.quat[1]; acados_in.x0[qy] = msg.quat[2]; acados_in.x0[qz] = msg.quat[3];
        acados_in.x0[vbx] = msg.vel[0]; acados_in.x0[vby] = msg.vel[1]; acados_in.x0[vbz] = msg.vel[2];
        acados_in.x0[wx] = msg.rates[0]; acados_in.x0[wy] = msg.rates[1]; acados_in.x0[wz] = msg.rates[2];
        // --- acados NMPC (:581-594)
        ocp_nlp_constraints_model_set(nlp_config, nlp_dims, nlp_in, 0, "lbx", acados_in.x0);
        ocp_nlp_constraints_model_set(nlp_config, nlp_dims, nlp_in, 0, "ubx", acados_in.x0);
        for (int i = 0; i < N; i++)
            for (int j = 0; j < NY; ++j) acados_in.yref[i * NY + j] = yref_sign[i * NY + j];
        for (int i = 0; i < NYN; i++) acados_in.yref_e[i] = yref_sign[N * NY + i];
        for (int ii = 0; ii < N; ii++) ocp_nlp_cost_model_set(nlp_config, nlp_dims, nlp_in, ii, "yref", acados_in.yref + ii * NY);
        ocp_nlp_cost_model_set(nlp_config, nlp_dims, nlp_in, N, "yref", acados_in.yref_e);
        // --- set constraints (:605-608, FIXED_U0): stage 0 is pinned to the input in flight, u1 of the previous step
        if (fixed_u0) {
            ocp_nlp_constraints_model_set(nlp_config, nlp_dims, nlp_in, 0, "lbu", acados_out.u1);
            ocp_nlp_constraints_model_set(nlp_config, nlp_dims, nlp_in, 0, "ubu", acados_out.u1);
        }
        // --- call solver (:611-616)
        acados_status = acados_solve();
        acados_out.status = acados_status;
        acados_out.KKT_res = (double)nlp_out->inf_norm_res;
        acados_out.cpu_time = (double)nlp_out->total_time;
        // --- get solution (:619-625): u0, u1, and x4 = stage 4 compensates the 60 ms delay
        ocp_nlp_out_get(nlp_config, nlp_dims, nlp_out, 0, "u", (void*)acados_out.u0);
        ocp_nlp_out_get(nlp_config, nlp_dims, nlp_out, 1, "u", (void*)acados_out.u1);
        ocp_nlp_out_get(nlp_config, nlp_dims, nlp_out, 4, "x", (void*)acados_out.x4);
        // --- motor speeds message (:628-642), double -> int32 truncation as on the wire
        last_motvel.w1 = (int32_t)acados_out.u0[w1]; last_motvel.w2 = (int32_t)acados_out.u0[w2];
        last_motvel.w3 = (int32_t)acados_out.u0[w3]; last_motvel.w4 = (int32_t)acados_out.u0[w4];
        // --- attitude / thrust command (:645-670)
        double qn[4] = {acados_out.x4[qw], acados_out.x4[qx], acados_out.x4[qy], acados_out.x4[qz]};
        const double nrm = std::sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
        for (double& c : qn) c /= nrm;
        const euler eu = attitude(qn);
        last_cmd_vel.linear_x = 1.0 * rad2Deg(eu.theta);
        last_cmd_vel.linear_y = -1.0 * rad2Deg(eu.phi);
        last_cmd_vel.linear_z = krpm2pwm((acados_out.u1[w1] + acados_out.u1[w2] + acados_out.u1[w3] + acados_out.u1[w4]) / 4);
        last_cmd_vel.angular_z = rad2Deg(acados_out.x4[wz]);
        return acados_status;
    }

private:
    const double* row(int r) const { return precomputed_traj.data() + (size_t)r * NY; }
    // one row of the Regulation / Position_Hold windows (acados_mpc.cpp:438-454, 497-513)
    void fill_hold_row(int k, double x, double y, double z) {
        double* r = yref_sign + k * NY;
        r[0] = x; r[1] = y; r[2] = z; r[3] = 1.00;
        for (int j = 4; j < 13; j++) r[j] = 0.00;
        for (int j = 13; j < 17; j++) r[j] = uss_row;
    }
};

}  // namespace cf
