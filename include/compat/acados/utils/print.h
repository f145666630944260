/* forwarding stub: the reference node includes this acados/generated header by name
 * (crazyflie_controller/src/acados_mpc.cpp:61-73); everything it needs is in one header. */
#include "acados_solver_crazyflie.h"
