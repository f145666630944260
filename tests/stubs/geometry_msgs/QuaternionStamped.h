#pragma once
#include <geometry_msgs/Vector3.h>
namespace geometry_msgs { struct QuaternionStamped { std_msgs::Header header; Quaternion quaternion; }; }
