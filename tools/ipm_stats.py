"""Development aid: distribution of (head, iterations) of the interior-point instances in the
bench workload, and how the interior-point kernel time scales with them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crazyflie_nmpc_amd import BatchSolver, default_opts, sim
from crazyflie_nmpc_amd.solver import INIT_HOVER
from crazyflie_nmpc_amd.synthetic import regulation_row, sample_hover_x0

B, N, KP = 65536, 50, 20
rng = np.random.default_rng(20200103)
dev = torch.device("cuda", 0)
x = torch.from_numpy(sample_hover_x0(rng, B)).to(dev)
row = regulation_row()
s = BatchSolver(B)
s.set_x0(x); s.set_yref(torch.from_numpy(np.tile(row, (B, N, 1))).to(dev), torch.from_numpy(np.tile(row[:13], (B, 1))).to(dev)); s.init_iterate(INIT_HOVER)
cohort = B // KP
kicks = torch.from_numpy(sample_hover_x0(rng, cohort * KP).reshape(KP, cohort, 13)).to(dev)
u0 = torch.empty((B, 4), dtype=torch.float64, device=dev); xn = torch.empty_like(x)
for t in range(30):
    x[(t % KP) * cohort:(t % KP + 1) * cohort].copy_(kicks[t % KP])
    s.set_x0(x); s.solve(1); s.get_u(0, out=u0); sim(x, u0, T=0.015, steps=1, out=xn); x, xn = xn, x
    if t >= 26:
        st, it, rs = s.stats(); hd = s.heads()
        m = it > 0
        print(f"step {t}: ipm instances {m.sum()}  iters mean {it[m].mean():.2f} max {it[m].max()}  heads {np.bincount(hd[m], minlength=51)[[4,8,12,16,24,32,50]]}")
        work = (it[m] * hd[m])
        idx = np.argsort(-work)[:8]
        print("   top (iters, head):", [(int(it[m][i]), int(hd[m][i])) for i in idx], " sum iters*head", int(work.sum()))
