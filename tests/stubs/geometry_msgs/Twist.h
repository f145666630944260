#pragma once
#include <geometry_msgs/Vector3.h>
namespace geometry_msgs { struct Twist { Vector3 linear, angular; }; }
