"""Development aid: where the interior-point kernel's LONGEST wave spends its time.  Needs a
library built with -DCFN_PROF (phase timers in k_ipm); pass its path as argv[1]:
    hipcc ... -DCFN_PROF -shared -o /tmp/libcfnmpc_prof.so -x hip cfnmpc_kernels.hip cfnmpc_api.cpp"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import crazyflie_nmpc_amd._lib as _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from crazyflie_nmpc_amd import BatchSolver, default_opts, sim
from crazyflie_nmpc_amd.solver import INIT_HOVER
from crazyflie_nmpc_amd.synthetic import regulation_row, sample_hover_x0
L = _lib.lib()
B, N, KP = int(os.environ.get("BATCH", "65536")), 50, 20
rng = np.random.default_rng(20200103)
dev = torch.device("cuda", 0)
SC = float(os.environ.get("KICK", "1"))
x = torch.from_numpy(sample_hover_x0(rng, B, scale=SC)).to(dev)
row = regulation_row()
s = BatchSolver(B, default_opts(active_set=int(os.environ.get("AS", "1"))))
s.set_x0(x); s.set_yref(torch.from_numpy(np.tile(row, (B, N, 1))).to(dev), torch.from_numpy(np.tile(row[:13], (B, 1))).to(dev)); s.init_iterate(INIT_HOVER)
cohort = B // KP
kicks = torch.from_numpy(sample_hover_x0(rng, cohort * KP, scale=SC).reshape(KP, cohort, 13)).to(dev)
u0 = torch.empty((B, 4), dtype=torch.float64, device=dev); xn = torch.empty_like(x)
names = ["gather", "elem(init,loop ctl)", "factor", "forward x2", "elem passes", "resolve", "rollout", "publish+commit"]
out = (C.c_ulonglong * 32)()
for t in range(30):
    x[(t % KP) * cohort:(t % KP + 1) * cohort].copy_(kicks[t % KP])
    s.set_x0(x)
    if t >= 26: L.cfnmpc_debug_prof(out, 1)
    s.solve(1); s.get_u(0, out=u0); sim(x, u0, T=0.015, steps=1, out=xn); x, xn = xn, x
    if t >= 26:
        L.cfnmpc_debug_prof(out, 0)
        v = np.array(list(out), dtype=np.float64) / 100.0  # wall_clock64: 100 MHz -> us
        print(f"step {t}: longest wave {v[8]:.0f} us, mean wave {v[9] / max(out[10], 1):.0f} us over {out[10]} waves")
        print(f"   longest wave: {out[11]} solves / interior-point iterations over {out[12]} stages; all waves: {out[13]} over {out[14]} stages; {out[15]} further attempts (longer head after a failed tail check)")
        print("   longest: " + "  ".join(f"{n} {v[i]:.0f}" for i, n in enumerate(names)))
        print("   mean   : " + "  ".join(f"{n} {v[16 + i] / max(out[10], 1):.0f}" for i, n in enumerate(names)))
