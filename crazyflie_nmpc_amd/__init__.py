"""crazyflie_nmpc_amd -- MI355X-native batched solve engine for the Crazyflie SQP-RTI hot path.

The product is `libcfnmpc.so` (hand-written HIP for gfx950 behind a C-ABI, include/cfnmpc.h and
include/acados_solver_crazyflie.h).  This package is the thin Python host side used by the
tests and bench.py: a ctypes binding (`_lib`), a batch solver object (`solver.BatchSolver`)
and a ROS-free mirror of the reference node's per-step protocol (`node`).
There is no CPU fallback: importing `_lib` fails loudly if the HIP library is missing.
"""
from .solver import BatchSolver, Opts, default_opts, estimate, sim  # noqa: F401

__all__ = ["BatchSolver", "Opts", "default_opts", "estimate", "sim"]
