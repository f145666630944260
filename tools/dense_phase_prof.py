"""Development aid: where k_as_dense's LONGEST row spends its time.  Needs a library built with -DCFN_PROF; its path is argv[1]:
    hipcc ... -DCFN_PROF -shared -o crazyflie_nmpc_amd/libcfnmpc_prof.so -x hip cfnmpc_kernels.hip cfnmpc_linfactor.hip cfnmpc_asdense.hip cfnmpc_pcond.hip cfnmpc_api.cpp cfnmpc_fleet.cpp cfnmpc_multi.cpp"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import crazyflie_nmpc_amd._lib as _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from crazyflie_nmpc_amd import BatchSolver, default_opts, sim
from crazyflie_nmpc_amd.solver import INIT_HOVER
from crazyflie_nmpc_amd.synthetic import regulation_row, sample_hover_x0
L = _lib.lib()
B, N, KP = int(os.environ.get("BATCH", "8192")), 50, 20
rng = np.random.default_rng(20200103)
dev = torch.device("cuda", 0)
x = torch.from_numpy(sample_hover_x0(rng, B)).to(dev)
row = regulation_row()
s = BatchSolver(B, default_opts(as_dense=1))
s.set_x0(x); s.set_yref(torch.from_numpy(np.tile(row, (B, N, 1))).to(dev), torch.from_numpy(np.tile(row[:13], (B, 1))).to(dev)); s.init_iterate(INIT_HOVER)
cohort = B // KP
kicks = torch.from_numpy(sample_hover_x0(rng, cohort * KP).reshape(KP, cohort, 13)).to(dev)
u0 = torch.empty((B, 4), dtype=torch.float64, device=dev); xn = torch.empty_like(x)
names = ["stage", "build", "invert+solves", "dx+publish"]
out = (C.c_ulonglong * 32)()
for t in range(30):
    x[(t % KP) * cohort:(t % KP + 1) * cohort].copy_(kicks[t % KP])
    s.set_x0(x)
    if t >= 26: L.cfnmpc_debug_dprof(out, 1)
    s.solve(1); s.get_u(0, out=u0); sim(x, u0, T=0.015, steps=1, out=xn); x, xn = xn, x
    if t >= 26:
        L.cfnmpc_debug_dprof(out, 0)
        v = np.array(list(out), dtype=np.float64) / 100.0  # wall_clock64: 100 MHz -> us
        n = max(out[10], 1)
        print(f"step {t}: {out[10]} rows; longest row {v[8]:.1f} us (head {out[12]}, {out[11]} solves), mean row {v[9] / n:.1f} us, mean solves {out[13] / n:.2f}")
        print("   longest: " + "  ".join(f"{nm} {v[i]:.1f}" for i, nm in enumerate(names)))
        print("   mean   : " + "  ".join(f"{nm} {v[16 + i] / n:.1f}" for i, nm in enumerate(names)))
