"""Development aid: of the rows that end in the interior-point fall-back, how many were sent there WITHOUT an active-set attempt
(unconstrained minimiser more than as_skip_viol box widths outside: known before k_as starts)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
import bench
from crazyflie_nmpc_amd import default_opts
dev = torch.device("cuda", 0)
for kick in (2.0, 3.0):
    f = bench.Fleet(65536, dev, np.random.default_rng(3), "hover", kick)
    for t in range(25): f.step()
    tot = skip = cons = 0
    for t in range(10):
        f.step()
        torch.cuda.synchronize()
        st, it, _ = f.solver.stats()
        v = np.empty(65536); f.solver._L.cfnmpc_debug_get_viol(f.solver._h, v.ctypes.data_as(C.c_void_p))
        fb = it > 12
        tot += fb.sum(); skip += (fb & (v > 4 * 22.0)).sum(); cons += (it > 0).sum()
    print(f"kick x{kick}: per step constrained {cons / 10:.0f}, fall-back rows {tot / 10:.1f}, of them beyond 4 widths (no active-set attempt) {skip / 10:.1f}")
    f.close()
