#pragma once
#include <geometry_msgs/Vector3.h>
namespace geometry_msgs { struct Vector3Stamped { std_msgs::Header header; Vector3 vector; }; typedef std::shared_ptr<Vector3Stamped> Vector3StampedPtr; typedef std::shared_ptr<const Vector3Stamped> Vector3StampedConstPtr; }
