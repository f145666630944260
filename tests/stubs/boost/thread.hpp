#pragma once
#include <functional>
namespace boost { using std::bind; }
using namespace std::placeholders;   // the nodes write boost::bind(..., _1, _2)
