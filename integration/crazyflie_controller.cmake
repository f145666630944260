# crazyflie_controller.cmake -- builds the reference's cf_nmpc / cf_estimator nodes against the
# MI355X-native drop-in instead of acados + the generated solver.
#
# Use from crazyflie_controller/CMakeLists.txt, AFTER catkin_package(...) and INSTEAD of the
# reference's acados block (CMakeLists.txt:17-20 `acados_include` / `acados_lib` /
# `full_model_build`, :53-65 include_directories / link_directories, :67-85 the two targets):
#
#     set(CFNMPC_ROOT "/path/to/this/repository")          # holds include/ and crazyflie_nmpc_amd/
#     include(${CFNMPC_ROOT}/integration/crazyflie_controller.cmake)
#
# The node sources stay untouched: include/compat/ carries forwarding headers under the names
# acados_mpc.cpp:61-73 and acados_estimator.cpp:66-76 include, and the library's weak definitions of
# the acados globals yield to the node's own (acados_mpc.cpp:76-84).
#
# Checked in the build container by tests/test_node_surface.py (both node sources parse against these
# include directories; every acados symbol their objects need is exported by the library).  NOT
# checked: a catkin build (no ROS in the build container or on the GPU box) -- SURVEY.md row N4.

if(NOT DEFINED CFNMPC_ROOT)
  message(FATAL_ERROR "set CFNMPC_ROOT to the root of the cfnmpc repository before including this file")
endif()

set(CFNMPC_DROPIN_LIB ${CFNMPC_ROOT}/crazyflie_nmpc_amd/libacados_solver_crazyflie.so)
set(CFNMPC_ENGINE_LIB ${CFNMPC_ROOT}/crazyflie_nmpc_amd/libcfnmpc.so)
if(NOT EXISTS ${CFNMPC_DROPIN_LIB} OR NOT EXISTS ${CFNMPC_ENGINE_LIB})
  message(FATAL_ERROR "build the libraries first: make -C ${CFNMPC_ROOT}/crazyflie_nmpc_amd/csrc  (needs hipcc, ROCm >= 7)")
endif()

include_directories(
  ${catkin_INCLUDE_DIRS}
  ${CFNMPC_ROOT}/include          # acados_solver_crazyflie.h, acados_sim_solver_crazyflie.h
  ${CFNMPC_ROOT}/include/compat   # acados/..., acados_c/..., blasfeo/include/..., crazyflie_model/... by name
)

# NMPC node (reference CMakeLists.txt:67-75)
add_executable(acados_mpc src/acados_mpc.cpp)
target_link_libraries(acados_mpc ${CFNMPC_DROPIN_LIB} ${CFNMPC_ENGINE_LIB} ${catkin_LIBRARIES})
add_dependencies(acados_mpc ${PROJECT_NAME}_gencfg ${PROJECT_NAME}_gencpp)

# Estimator node (reference CMakeLists.txt:77-85)
add_executable(acados_estimator src/acados_estimator.cpp)
target_link_libraries(acados_estimator ${CFNMPC_DROPIN_LIB} ${CFNMPC_ENGINE_LIB} ${catkin_LIBRARIES})
add_dependencies(acados_estimator ${PROJECT_NAME}_gencfg ${PROJECT_NAME}_gencpp)

# both libraries locate each other and the ROCm runtime through their own RPATH / the loader path;
# at run time the nodes need a HIP device (acados_create() returns non-zero otherwise and the node
# exits as the reference does, acados_mpc.cpp:227-230)
set_target_properties(acados_mpc acados_estimator PROPERTIES BUILD_RPATH "${CFNMPC_ROOT}/crazyflie_nmpc_amd"
                      INSTALL_RPATH "${CFNMPC_ROOT}/crazyflie_nmpc_amd")
