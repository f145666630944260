#pragma once
#include <geometry_msgs/Vector3.h>
