#pragma once
#include <geometry_msgs/Vector3.h>
namespace sensor_msgs {
struct Imu {
    typedef std::shared_ptr<Imu> Ptr;
    typedef std::shared_ptr<const Imu> ConstPtr;
    std_msgs::Header header; geometry_msgs::Quaternion orientation; geometry_msgs::Vector3 angular_velocity, linear_acceleration;
};
typedef Imu::Ptr ImuPtr; typedef Imu::ConstPtr ImuConstPtr;
}
