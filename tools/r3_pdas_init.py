"""Round-3 numpy experiment (negative): initial active-set guess from a SATURATED roll-out of the unconstrained feedback law
instead of the violations of the unconstrained minimiser -- mean solves 1.52 -> 1.35 / 2.27 -> 2.00, the tail unchanged, and the
extra forward sweep costs what it saves."""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from oracle import cfnmpc_oracle as o
N = 50
yref, yref_e = o.regulation_yref(N, (0, 0, 0.4))
hov = np.array([0, 0, 0.4, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0.0])

def pdas(H, h, lb, ub, lo, up, maxs=30):
    for s in range(1, maxs + 1):
        act = lo | up; free = ~act
        v = np.where(lo, lb, np.where(up, ub, 0.0))
        if free.any():
            v[free] = np.linalg.solve(H[np.ix_(free, free)], -h[free] - H[np.ix_(free, act)] @ v[act])
        grad = H @ v + h
        lo2 = (free & (v < lb)) | (lo & (grad > 0)); up2 = (free & (v > ub)) | (up & (grad < 0))
        if np.array_equal(lo2, lo) and np.array_equal(up2, up): return s
        lo, up = lo2, up2
    return 99

def sat_rollout_guess(qp):
    """forward sweep of the unconstrained feedback law with the inputs clipped to the box"""
    Rd = np.tile(qp.Rd, (qp.N, 1))
    K, Sinv, d = o._riccati_factor(qp, Rd, qp.r, absolute=True)
    x = qp.dx0.copy()
    lo = np.zeros((qp.N, 4), bool); up = np.zeros((qp.N, 4), bool)
    for k in range(qp.N):
        v = -K[k] @ x - d[k]
        lo[k] = v < qp.lb[k]; up[k] = v > qp.ub[k]
        v = np.clip(v, qp.lb[k], qp.ub[k])
        x = qp.A[k] @ x + qp.B[k] @ v + qp.b[k]
    return lo.reshape(-1), up.reshape(-1)

for scale in (1.0, 2.0):
    rng = np.random.default_rng(11)
    res = {'viol': [], 'sat': [], 'union': []}
    for x0 in o.sample_hover_x0(rng, 250, scale=scale):
        xbar = np.tile(hov, (N + 1, 1)); ubar = np.full((N, 4), o.HOV_W)
        qp = o.build_qp(xbar, ubar, x0, yref, yref_e, jac=o.jac_fd)
        H, h, _, _ = o.condense(qp)
        lb, ub = qp.lb.reshape(-1), qp.ub.reshape(-1)
        v0 = np.linalg.solve(H, -h)
        lo0, up0 = v0 < lb, v0 > ub
        if not (lo0.any() or up0.any()): continue
        res['viol'].append(pdas(H, h, lb, ub, lo0, up0))
        lo1, up1 = sat_rollout_guess(qp)
        res['sat'].append(1 + pdas(H, h, lb, ub, lo1, up1) if (lo1.any() or up1.any()) else 1)   # +1: the extra forward sweep ~ half a solve; count as shown below
        res['union'].append(pdas(H, h, lb, ub, lo0 | lo1, up0 | up1))
    for k, v in res.items():
        v = np.array(v)
        extra = 1 if k == 'sat' else 0
        vv = v - extra
        print(f"scale {scale} {k:6s}: n {len(v)} mean solves {vv.mean():.2f} hist {np.bincount(np.minimum(vv, 12))[:13]} max {vv.max()}")
