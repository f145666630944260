"""Development aid: backward half of the start solve, two kernels (k_linearise + k_factor) against the fused k_linfactor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch
import cfnmpc_oracle as o
from crazyflie_nmpc_amd import BatchSolver
from crazyflie_nmpc_amd.solver import INIT_HOVER

for B in [int(a) for a in (sys.argv[1:] or ["4096", "16384", "65536"])]:
    rng = np.random.default_rng(1)
    x0 = o.sample_hover_x0(rng, B)
    yr, ye = o.regulation_yref(50, (0, 0, 0.4))
    yref = np.repeat(yr[None], B, 0).copy(); yref_e = np.repeat(ye[None], B, 0).copy()
    s = BatchSolver(B)
    s.set_x0(x0); s.set_yref(yref, yref_e); s.init_iterate(INIT_HOVER)
    s.solve(3)
    torch.cuda.synchronize()
    for rep in range(3):
        t1 = s.start_factor(1, 20); t2 = s.start_factor(2, 20)
        print(f"B {B}: k_linearise + k_factor {t1:.4f} ms   k_linfactor {t2:.4f} ms", flush=True)
    s.close()
