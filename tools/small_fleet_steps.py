"""Development aid: per step of a small fleet's closed loop -- the six kernel-group durations, the work-list counts and the head
classes of the constrained rows (which steps pay for a row with a long head / a retry).   python tools/small_fleet_steps.py [batch] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crazyflie_nmpc_amd import BatchSolver, default_opts, sim
from crazyflie_nmpc_amd.solver import INIT_HOVER
from crazyflie_nmpc_amd.synthetic import regulation_row, sample_hover_x0

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
N, KP = 50, 20
rng = np.random.default_rng(20200107)
dev = torch.device("cuda", 0)
x = torch.from_numpy(sample_hover_x0(rng, B)).to(dev)
row = regulation_row()
s = BatchSolver(B)
s.set_x0(x); s.set_yref(torch.from_numpy(np.tile(row, (B, N, 1))).to(dev), torch.from_numpy(np.tile(row[:13], (B, 1))).to(dev)); s.init_iterate(INIT_HOVER)
cohort = B // KP
kicks = torch.from_numpy(sample_hover_x0(rng, cohort * KP).reshape(KP, cohort, 13)).to(dev)
u0 = torch.empty((B, 4), dtype=torch.float64, device=dev); xn = torch.empty_like(x)
rows = []
for t in range(20 + steps):
    x[(t % KP) * cohort:(t % KP + 1) * cohort].copy_(kicks[t % KP])
    if t == 20:
        s.set_profiling(True)
    s.set_x0(x); s.solve(1); s.get_u(0, out=u0); sim(x, u0, T=0.015, steps=1, out=xn); x, xn = xn, x
    if t >= 20:
        torch.cuda.synchronize()
        st, it, rs = s.stats(); hd = s.heads(); cnt = s.list_counts()
        m = it > 0
        rows.append((cnt, np.bincount(hd[m], minlength=51)[[4, 8, 12, 16, 24, 32, 50]], int(it.max()), int((rs > 0).sum())))
per = s.get_profile_steps()
for (cnt, hb, itmax, nip), p in zip(rows, per):
    print(f"AS group {p[4] + p[5]:.3f} ms (step kernels {p.sum():.3f}) | listed {cnt[0]} long-head {cnt[2]} late {cnt[3]} | final heads 4..N {hb.tolist()} | max solves {itmax} | interior-point rows {nip}")
print("mean step kernels", per.sum(axis=1).mean(), "AS group mean", (per[:, 4] + per[:, 5]).mean())
