"""Round-6 numpy experiment: would the interior-point fall-back of a row whose active set did NOT settle within 12 solves
need fewer iterations when it starts from the active-set iteration's last iterate (clipped into the box) instead of from the
unconstrained minimiser?  Dense Mehrotra iteration with the engine's rules (tools/r3_ipm_start.py), QPs = first RTI step from the
hover iterate for vehicles kicked at `scale` x the bench's disturbance.
    python tools/r6_ipm_warm.py [n] [scale]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cfnmpc_oracle as o
N = 50
yref, yref_e = o.regulation_yref(N, (0, 0, 0.4))


def steplen(tl, tu, ll, lu, dtl, dtu, dll, dlu):
    a = 1.0
    for z, dz in ((tl, dtl), (tu, dtu), (ll, dll), (lu, dlu)):
        m = dz < 0
        if m.any(): a = min(a, float((-z[m] / dz[m]).min()))
    return a


def ipm(H, h, lb, ub, vstart=None, m=0.05, tol=1e-8, max_iter=80, thr0=1.0, lam0_min=1e-2, mu0_scale=0.1, tau=0.995, clip_viol=2.0):
    """vstart None: the engine's rule (infeasible start, clipped start beyond clip_viol widths); else: clipped start from vstart"""
    n = len(h); nc = 2 * n
    v0 = np.linalg.solve(H, -h)
    w = ub - lb
    viol = max(np.maximum(lb - v0, 0).max(), np.maximum(v0 - ub, 0).max())
    if vstart is None and viol <= clip_viol * w.max():
        v = v0.copy()
        tl = np.maximum(v - lb, thr0); tu = np.maximum(ub - v, thr0)
        mu0 = max(lam0_min, mu0_scale * viol)
        ll = mu0 / tl; lu = mu0 / tu
        rg = -ll + lu
    else:
        v = np.clip(v0 if vstart is None else vstart, lb + m * w, ub - m * w)
        g = H @ v + h
        tl = v - lb; tu = ub - v
        mu0 = max(lam0_min, mu0_scale * float((np.abs(g) * np.minimum(tl, tu)).mean()))
        ll = np.maximum(g, 0) + mu0 / tl; lu = np.maximum(-g, 0) + mu0 / tu
        rg = g - ll + lu
    it = 0
    while True:
        rl = v - lb - tl; ru = ub - v - tu
        mu = float((ll * tl).sum() + (lu * tu).sum()) / nc
        res = max((ll * tl).max(), (lu * tu).max(), np.abs(rg).max(), np.abs(rl).max(), np.abs(ru).max())
        if res <= tol: return it, v
        if it >= max_iter: return -it, v
        it += 1
        Dl, Du = ll / tl, lu / tu
        c = np.linalg.cholesky(H + np.diag(Dl + Du))
        solve = lambda r: np.linalg.solve(c.T, np.linalg.solve(c, r))
        g_aff = rg + ll + Dl * rl - lu - Du * ru
        dv_a = solve(-g_aff)
        dtl_a = dv_a + rl; dtu_a = -dv_a + ru
        dll_a = -ll - Dl * dtl_a; dlu_a = -lu - Du * dtu_a
        a_aff = steplen(tl, tu, ll, lu, dtl_a, dtu_a, dll_a, dlu_a)
        mu_aff = float(((ll + a_aff * dll_a) * (tl + a_aff * dtl_a)).sum() + ((lu + a_aff * dlu_a) * (tu + a_aff * dtu_a)).sum()) / nc
        smu = (mu_aff / mu) ** 3 * mu
        cl = dll_a * dtl_a; cu = dlu_a * dtu_a
        dv = dv_a + solve(-((cl - smu) / tl - (cu - smu) / tu))
        dtl = dv + rl; dtu = -dv + ru
        dll = (smu - cl) / tl - ll - Dl * dtl; dlu = (smu - cu) / tu - lu - Du * dtu
        a = min(1.0, tau * steplen(tl, tu, ll, lu, dtl, dtu, dll, dlu))
        v = v + a * dv; tl, tu = tl + a * dtl, tu + a * dtu; ll, lu = ll + a * dll, lu + a * dlu
        rg = (1.0 - a) * rg


n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
# QPs whose active set does not settle come from the CLOSED LOOP (saturated iterates meeting new states), not from the hover
# iterate: run the C restatement's loop with staggered kicks and capture the rows that fell back without having been skipped
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import cref
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
rows = []
nuns = 0
for xp, up, x0, it_c in cap[:1500]:
    qp = o.build_qp(xp, up, x0, yref, yref_e, jac=o.jac_fd)
    H, h, _, _ = o.condense(qp)
    lb, ub = qp.lb.reshape(-1), qp.ub.reshape(-1)
    v0 = np.linalg.solve(H, -h)
    viol = max(np.maximum(lb - v0, 0).max(), np.maximum(v0 - ub, 0).max()) / 22.0
    if viol > 4.0: continue          # skipped the active set (as_skip_viol): not the rows in question
    pd = o.pdas_dense(qp, max_solves=12)
    if pd["converged"]: continue
    nuns += 1
    try:
        base, vb = ipm(H, h, lb, ub)
        res = [base]
        for m in (0.05, 0.02, 0.005):
            it, v = ipm(H, h, lb, ub, vstart=pd["du"].reshape(-1), m=m)
            res.append(it)
    except np.linalg.LinAlgError:
        continue
    rows.append([viol, it_c] + res)
rows = np.array(rows)
print(f"{nuns} of the first 1500 with an unsettled active set after 12 solves and a violation below 4 widths; violation median {np.median(rows[:, 0]):.2f} widths; "
      f"restatement's own iterations median {np.median(rows[:, 1]):.0f}")
for j, name in enumerate(("engine rule (start from the unconstrained minimiser)", "from the last active-set iterate, margin 5 %", "margin 2 %", "margin 0.5 %")):
    c = rows[:, 2 + j]
    q = np.percentile(np.abs(c), [50, 90, 99, 100])
    print(f"{name:55s}: iterations mean {np.abs(c).mean():.1f} p50 / p90 / p99 / max {q} failed {(c < 0).sum()}")
