"""Round-6 numpy experiment: the rows of the closed loop whose active set is NOT stationary after 12 solves (two thirds of the
interior-point fall-back under heavy disturbances, profiles/r06_notes.md section 4) -- how many solves would they need with a
higher cap, and do they cycle?  Rows captured from the C restatement's closed loop with staggered kicks (as tools/r6_ipm_warm.py).
    python tools/r6_as_cap.py [n] [scale]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from oracle import cfnmpc_oracle as o
import cref
N = 50
yref, yref_e = o.regulation_yref(N, (0, 0, 0.4))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
cref.build()
rng = np.random.default_rng(3)
B = n
x = o.sample_hover_x0(rng, B, scale=scale)
yr = np.repeat(yref[None], B, 0).copy(); ye = np.repeat(yref_e[None], B, 0).copy()
xr = np.repeat(x[:, None, :], N + 1, 1).copy(); ur = np.full((B, N, 4), o.HOV_W)
opts = cref.default_opts(active_set=1)
cap = []
KP = 20
cohort = B // KP
for t in range(40):
    c0 = (t % KP) * cohort
    x[c0:c0 + cohort] = o.sample_hover_x0(rng, cohort, scale=scale)
    xp, up = xr.copy(), ur.copy()
    st, it, rs, _ = cref.rti_step(opts, xr, ur, x.copy(), yr, ye, nthreads=0)
    fb = np.nonzero((rs > 0) & (st == 0))[0]
    for i in fb:
        cap.append((xp[i].copy(), up[i].copy(), x[i].copy(), int(it[i])))
    x = cref.sim(x, ur[:, 0, :].copy(), T=0.015, steps=1)
print(f"captured {len(cap)} fall-back rows from {B} vehicles x 40 steps at kick scale {scale}")


def pdas_trace(H, h, lb, ub, max_solves=200):
    v0 = np.linalg.solve(H, -h)
    lo, up = v0 < lb, v0 > ub
    seen = {}
    for s in range(1, max_solves + 1):
        act = lo | up; free = ~act
        v = np.where(lo, lb, np.where(up, ub, 0.0))
        if free.any():
            v[free] = np.linalg.solve(H[np.ix_(free, free)], -h[free] - H[np.ix_(free, act)] @ v[act])
        grad = H @ v + h
        lo2 = (free & (v < lb)) | (lo & (grad > 0))
        up2 = (free & (v > ub)) | (up & (grad < 0))
        if np.array_equal(lo2, lo) and np.array_equal(up2, up):
            return s, 0, int(act.sum())
        key = (lo2.tobytes(), up2.tobytes())
        if key in seen:
            return -s, s - seen[key], int(act.sum())      # cycle of that period
        seen[key] = s
        lo, up = lo2, up2
    return -max_solves, 0, int(act.sum())


res = []
for xp, up_, x0, it_c in cap[:2500]:
    qp = o.build_qp(xp, up_, x0, yref, yref_e, jac=o.jac_fd)
    H, h, _, _ = o.condense(qp)
    lb, ub = qp.lb.reshape(-1), qp.ub.reshape(-1)
    v0 = np.linalg.solve(H, -h)
    viol = max(np.maximum(lb - v0, 0).max(), np.maximum(v0 - ub, 0).max()) / 22.0
    s, per, nact = pdas_trace(H, h, lb, ub)
    res.append((viol, s, per, nact, it_c))
res = np.array(res, dtype=float)
skip = res[:, 0] > 4.0
print(f"{len(res)} rows; {skip.sum()} skipped the active set (violation > 4 widths)")
for name, m in (("violation <= 4 widths (tried the active set)", ~skip), ("violation > 4 widths (skipped it)", skip)):
    r = res[m]
    if not len(r): continue
    s = r[:, 1]
    conv = s > 0
    print(f"{name}: {len(r)} rows; settle at any count: {conv.sum()} ({100 * conv.mean():.0f} %); cycle: {(~conv).sum()} (periods {np.unique(r[~conv, 2]).astype(int).tolist()}, detected after median {np.median(-s[~conv]) if (~conv).any() else 0:.0f} solves)")
    for capv in (12, 14, 16, 20, 24, 32, 48, 64):
        print(f"   cap {capv:3d}: settled {(conv & (s <= capv)).sum():5d} = {100 * (conv & (s <= capv)).mean():5.1f} %")
    print(f"   solves of the settling rows: p50 / p90 / p99 / max = {np.percentile(s[conv], [50, 90, 99, 100]) if conv.any() else None}; restatement's interior-point iterations p50 / max {np.percentile(r[:, 4], [50, 100])}")
