// declaration-only stand-in (see tests/stubs/README.md)
#pragma once
#include <cstdint>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>
namespace ros {
struct Time { double t = 0; static Time now() { return Time(); } double toSec() const { return t; } bool isZero() const { return t == 0; } };
struct Duration { explicit Duration(double) {} };
struct TimerEvent { Time last_real, current_real; };
struct Timer {};
struct Publisher { template <class M> void publish(const M&) const {} };
struct Subscriber {};
struct NodeHandle {
    NodeHandle() {}
    explicit NodeHandle(const std::string&) {}
    template <class M> Publisher advertise(const std::string&, int) { return Publisher(); }
    template <class M, class T> Subscriber subscribe(const std::string&, int, void (T::*)(const M&), T*) { return Subscriber(); }
    template <class T> Timer createTimer(Duration, void (T::*)(const TimerEvent&), T*) { return Timer(); }
    template <class V> bool getParam(const std::string&, V&) const { return true; }
    template <class V, class D> void param(const std::string&, V&, const D&) const {}
};
inline void init(int&, char**, const std::string&) {}
inline void spin() {}
}  // namespace ros
#define ROS_INFO(...) ((void)0)
#define ROS_WARN(...) ((void)0)
#define ROS_DEBUG(...) ((void)0)
#define ROS_ERROR(...) ((void)0)
#define ROS_INFO_STREAM(x) do { std::ostringstream ros_stub_os_; ros_stub_os_ << x; } while (0)
#define ROS_WARN_STREAM(x) ROS_INFO_STREAM(x)
namespace std_msgs { struct Header { uint32_t seq = 0; ros::Time stamp; std::string frame_id; }; }
