#pragma once
#include <geometry_msgs/Vector3.h>
namespace geometry_msgs { struct PointStamped { std_msgs::Header header; Point point; }; typedef std::shared_ptr<PointStamped> PointStampedPtr; typedef std::shared_ptr<const PointStamped> PointStampedConstPtr; }
