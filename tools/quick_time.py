"""Quick GPU timing of the solver kernels (development aid, not the bench contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch
import cfnmpc_oracle as o
from crazyflie_nmpc_amd import BatchSolver
from crazyflie_nmpc_amd.solver import INIT_HOVER

for B in (4096, 65536):
    rng = np.random.default_rng(1)
    x0 = o.sample_hover_x0(rng, B)
    yr, ye = o.regulation_yref(50, (0, 0, 0.4))
    yref = np.repeat(yr[None], B, 0).copy(); yref_e = np.repeat(ye[None], B, 0).copy()
    s = BatchSolver(B)
    print("B", B, "workspace GB", s.workspace_bytes / 1e9)
    s.set_x0(x0); s.set_yref(yref, yref_e); s.init_iterate(INIT_HOVER)
    for t in range(6):
        torch.cuda.synchronize(); t0 = time.time()
        s.solve(1)
        torch.cuda.synchronize(); dt = time.time() - t0
        st, it, rs = s.stats()
        hd = s.heads()
        print(f"  step {t}: {dt*1e3:.2f} ms  {B/dt/1e6:.3f} Msteps/s  iters mean {it.mean():.2f} max {it.max()} frac>0 {(it>0).mean():.2f} status {np.bincount(st)} heads {np.bincount(hd, minlength=51)[[4,8,12,16,24,32,50]]}")
    t0 = time.time(); s.linearise_only(); torch.cuda.synchronize(); print("  linearise only: %.2f ms" % ((time.time() - t0) * 1e3))
    s.close()
