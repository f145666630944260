"""Development aid: where a heavily disturbed fleet's step goes (kicks x 2 / x 3, the sensitivity entries that miss the
reference's 15 ms period): per-kernel split, work-list counts, and the (head, iterations) distribution of the rows the
interior-point fall-back ends up with.    python tools/kick_stats.py [kick_scale] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crazyflie_nmpc_amd import BatchSolver, default_opts, sim
from crazyflie_nmpc_amd.solver import INIT_HOVER
from crazyflie_nmpc_amd.synthetic import regulation_row, sample_hover_x0

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
N, KP = 50, 20
rng = np.random.default_rng(20200103)
dev = torch.device("cuda", 0)
x = torch.from_numpy(sample_hover_x0(rng, B, scale=scale)).to(dev)
row = regulation_row()
s = BatchSolver(B)
s.set_x0(x); s.set_yref(torch.from_numpy(np.tile(row, (B, N, 1))).to(dev), torch.from_numpy(np.tile(row[:13], (B, 1))).to(dev)); s.init_iterate(INIT_HOVER)
cohort = B // KP
kicks = torch.from_numpy(sample_hover_x0(rng, cohort * KP, scale=scale).reshape(KP, cohort, 13)).to(dev)
u0 = torch.empty((B, 4), dtype=torch.float64, device=dev); xn = torch.empty_like(x)
for t in range(44):
    x[(t % KP) * cohort:(t % KP + 1) * cohort].copy_(kicks[t % KP])
    if t == 30:
        s.set_profiling(True)
    s.set_x0(x); s.solve(1); s.get_u(0, out=u0); sim(x, u0, T=0.015, steps=1, out=xn); x, xn = xn, x
    if t >= 40:
        st, it, rs = s.stats(); hd = s.heads(); cnt = s.list_counts()
        m = it > 0
        ipm = it > 12            # (more than the active-set cap: certainly interior-point iterations)
        print(f"step {t}: constrained {m.sum()} ({m.mean():.3f}), list counts {cnt}, status != 0: {(st != 0).sum()}, it>12: {ipm.sum()}, "
              f"heads of all {np.bincount(hd[m], minlength=51)[[4,8,12,16,24,32,50]]}, of it>12 {np.bincount(hd[ipm], minlength=51)[[4,8,12,16,24,32,50]]}")
        viol = np.empty(B); s._L.cfnmpc_debug_get_viol(s._h, viol.ctypes.data_as(__import__('ctypes').c_void_p))
        skip = viol > 4.0 * 22.0
        fb = m & (rs > 0)          # interior-point rows: residual > 0 (active-set rows report exactly 0)
        print(f"   skip rows (viol > 4 widths) {skip.sum()}, interior-point rows {fb.sum()} of which skipped {(fb & skip).sum()}, failed active set {(fb & ~skip).sum()}; "
              f"active-set solves histogram of settled rows {np.bincount(it[m & ~fb], minlength=13)[1:13]}; ipm iterations of skip rows p50/max {np.percentile(it[fb & skip], [50, 100]) if (fb & skip).any() else None}, of failed rows {np.percentile(it[fb & ~skip], [50, 100]) if (fb & ~skip).any() else None}; viol quantiles of failed rows {np.percentile(viol[fb & ~skip], [10, 50, 90]) if (fb & ~skip).any() else None}")
        if ipm.any():
            q = np.percentile(it[ipm], [50, 90, 99, 100])
            print(f"   it>12 rows: iterations p50/p90/p99/max {q}, sum it*head {(it[ipm] * hd[ipm]).sum()}")
per = s.get_profile_steps()
names = ["linearise", "factor", "forward", "compaction", "active set", "interior point"]
print("per-kernel ms (mean | max over", len(per), "steps):", {n: (round(float(per[:, j].mean()), 3), round(float(per[:, j].max()), 3)) for j, n in enumerate(names)})
