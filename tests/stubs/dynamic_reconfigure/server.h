#pragma once
#include <cstdint>
#include <functional>
namespace dynamic_reconfigure {
template <class C> struct Server { typedef std::function<void(C&, uint32_t)> CallbackType; void setCallback(const CallbackType&) {} };
}
