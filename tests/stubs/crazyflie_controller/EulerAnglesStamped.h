#pragma once
#include <crazyflie_controller/msgs.h>
