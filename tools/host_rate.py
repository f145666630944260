"""PCIe-inclusive rate of the batch boundary with HOST buffers: closed-loop hover
fleet where every step hands x0 over from host memory and takes u0 / u1 / x4 back to host memory
(what NMPC::iteration exchanges with the solver, acados_mpc.cpp:581-625), (a) with the references
resident (regulation: yref does not change) and (b) with yref [B][N][17] handed over every step as well
(tracking through host windows).  The plant runs on the host side of the boundary only as a device
call with host pointers.  Not the bench's `value`: bench.py times device-resident inputs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crazyflie_nmpc_amd import BatchSolver, default_opts, sim
from crazyflie_nmpc_amd.solver import INIT_HOVER
from crazyflie_nmpc_amd.synthetic import regulation_row, sample_hover_x0

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N, K, W = 50, 20, 5
rng = np.random.default_rng(20200103)
x = sample_hover_x0(rng, B)
row = regulation_row()
yref = np.ascontiguousarray(np.tile(row, (B, N, 1))); yref_e = np.ascontiguousarray(np.tile(row[:13], (B, 1)))
s = BatchSolver(B, default_opts())
s.set_x0(x); s.set_yref(yref, yref_e); s.init_iterate(INIT_HOVER)
u0 = np.empty((B, 4)); u1 = np.empty((B, 4)); x4 = np.empty((B, 13)); xn = np.empty_like(x)
for with_yref in (0, 1):
    for t in range(W + K):
        if t == W:
            t0 = time.perf_counter()
        if with_yref:
            s.set_yref(yref, yref_e)
        s.set_x0(x); s.solve(1)
        s.get_u(0, out=u0); s.get_u(1, out=u1); s.get_x(4, out=x4)   # host copies: synchronous
        sim(x, u0, T=0.015, steps=1, out=xn); x, xn = xn, x
    dt = (time.perf_counter() - t0) / K
    print(f"B = {B}, host buffers, yref {'every step' if with_yref else 'resident'}: {dt * 1e3:.2f} ms per step, {B / dt / 1e6:.2f} M RTI steps/s")
