// message structs of crazyflie_controller/msg/*.msg, field for field (declaration-only stand-in)
#pragma once
#include <geometry_msgs/Vector3.h>
namespace crazyflie_controller {
struct CrazyflieState { geometry_msgs::Vector3 pos; geometry_msgs::Quaternion quat; geometry_msgs::Vector3 vel, rates; };
struct PropellerSpeeds { int32_t w1 = 0, w2 = 0, w3 = 0, w4 = 0; };
struct CrazyflieStateStamped { std_msgs::Header header; geometry_msgs::Vector3 pos; geometry_msgs::Quaternion quat; geometry_msgs::Vector3 vel, rates; };
struct PropellerSpeedsStamped { std_msgs::Header header; int32_t w1 = 0, w2 = 0, w3 = 0, w4 = 0; };
struct CrazyflieOpenloopTraj { std_msgs::Header header; double cpu_time = 0; std::vector<CrazyflieState> states; std::vector<PropellerSpeeds> controls; };
struct EulerAnglesStamped { std_msgs::Header header; double roll = 0, pitch = 0, yaw = 0; };
struct GenericLogData { std_msgs::Header header; std::vector<double> values; };
#define CF_STUB_PTRS(T) typedef std::shared_ptr<T> T##Ptr; typedef std::shared_ptr<const T> T##ConstPtr;
CF_STUB_PTRS(CrazyflieState) CF_STUB_PTRS(PropellerSpeeds) CF_STUB_PTRS(CrazyflieStateStamped) CF_STUB_PTRS(PropellerSpeedsStamped)
CF_STUB_PTRS(CrazyflieOpenloopTraj) CF_STUB_PTRS(EulerAnglesStamped) CF_STUB_PTRS(GenericLogData)
#undef CF_STUB_PTRS
// config/crazyflie_params.cfg:9-35
struct crazyflie_paramsConfig {
    bool enable_traj_tracking = false, enable_regulation = true;
    double xq_des = 0, yq_des = 0, zq_des = 0.40;
    double Wdiag_xq, Wdiag_yq, Wdiag_zq, Wdiag_qw, Wdiag_qx, Wdiag_qy, Wdiag_qz, Wdiag_vbx, Wdiag_vby, Wdiag_vbz, Wdiag_wx,
        Wdiag_wy, Wdiag_wz, Wdiag_w1, Wdiag_w2, Wdiag_w3, Wdiag_w4;
};
// config/crazyflie_estimator.cfg:9
struct crazyflie_estimatorConfig { double delay = 0.015; };
}  // namespace crazyflie_controller
