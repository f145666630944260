#pragma once
#include <ros/ros.h>
namespace std_msgs { struct String { std::string data; }; }
