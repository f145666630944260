"""Round-3 numpy experiment behind cfnmpc_opts.ipm_clip_viol: dense Mehrotra iteration with the
engine's rules, start variants compared on ordinary and on captured fall-back QPs.
    python tools/r3_ipm_start.py hard|st2          (needs gpurun_out/hard_cases.npz / st2_cases.npz from tools/r3_capture_st2.py)
    python tools/r3_ipm_start.py typ <n> <kick scale>"""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from oracle import cfnmpc_oracle as o
N = 50
yref, yref_e = o.regulation_yref(N, (0, 0, 0.4))

def steplen(tl, tu, ll, lu, dtl, dtu, dll, dlu):
    a = 1.0
    for z, dz in ((tl, dtl), (tu, dtu), (ll, dll), (lu, dlu)):
        m = dz < 0
        if m.any(): a = min(a, float((-z[m] / dz[m]).min()))
    return a

def ipm(H, h, lb, ub, start, tol=1e-8, max_iter=80, thr0=1.0, lam0_min=1e-2, mu0_scale=0.1, tau=0.995, m=0.05):
    n = len(h); nc = 2 * n
    v0 = np.linalg.solve(H, -h)
    if np.all(v0 >= lb) and np.all(v0 <= ub): return 0, v0
    if start == 'current':
        v = v0.copy()
        tl = np.maximum(v - lb, thr0); tu = np.maximum(ub - v, thr0)
        viol = max(np.maximum(lb - v, 0).max(), np.maximum(v - ub, 0).max())
        mu0 = max(lam0_min, mu0_scale * viol)
        ll = mu0 / tl; lu = mu0 / tu
        rg = -ll + lu
    else:
        w = ub - lb
        v = np.clip(v0, lb + m * w, ub - m * w)
        g = H @ v + h
        tl = v - lb; tu = ub - v
        if start == 'clip_absorb':      # multipliers absorb the gradient + a floor
            mu0 = max(lam0_min, mu0_scale * np.abs(g * np.minimum(tl, tu)).mean()) if False else lam0_min
            ll = np.maximum(g, 0) + mu0 / tl; lu = np.maximum(-g, 0) + mu0 / tu
        elif start == 'clip_mu':        # uniform mu0 scaled by the gradient
            mu0 = max(lam0_min, float(np.abs(g).mean()) * float(np.minimum(tl, tu).mean()) * mu0_scale)
            ll = mu0 / tl; lu = mu0 / tu
        elif start == 'clip_absorb_mu':
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
        M = H + np.diag(Dl + Du)
        g_aff = rg + ll + Dl * rl - lu - Du * ru
        c = np.linalg.cholesky(M)
        solve = lambda r: np.linalg.solve(c.T, np.linalg.solve(c, r))
        dv_a = solve(-g_aff)
        dtl_a = dv_a + rl; dtu_a = -dv_a + ru
        dll_a = -ll - Dl * dtl_a; dlu_a = -lu - Du * dtu_a
        a_aff = steplen(tl, tu, ll, lu, dtl_a, dtu_a, dll_a, dlu_a)
        mu_aff = float(((ll + a_aff * dll_a) * (tl + a_aff * dtl_a)).sum() + ((lu + a_aff * dlu_a) * (tu + a_aff * dtu_a)).sum()) / nc
        smu = (mu_aff / mu) ** 3 * mu
        cl = dll_a * dtl_a; cu = dlu_a * dtu_a
        g_cor = (cl - smu) / tl - (cu - smu) / tu
        dv = dv_a + solve(-g_cor)
        dtl = dv + rl; dtu = -dv + ru
        dll = (smu - cl) / tl - ll - Dl * dtl; dlu = (smu - cu) / tu - lu - Du * dtu
        a = min(1.0, tau * steplen(tl, tu, ll, lu, dtl, dtu, dll, dlu))
        v = v + a * dv; tl, tu = tl + a * dtl, tu + a * dtu; ll, lu = ll + a * dll, lu + a * dlu
        rg = (1.0 - a) * rg

def qps_hard():
    d = np.load('gpurun_out/hard_cases.npz')
    for i in range(len(d['it'])):
        qp = o.build_qp(d['xit'][i], d['uit'][i], d['x0'][i], yref, yref_e, jac=o.jac_fd)
        H, h, _, _ = o.condense(qp)
        yield H, h, qp.lb.reshape(-1), qp.ub.reshape(-1), int(d['it'][i])

def qps_typ(n, scale, seed=1):
    rng = np.random.default_rng(seed)
    hov = np.array([0, 0, 0.4, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0.0])
    for x0 in o.sample_hover_x0(rng, n, scale=scale):
        xbar = np.tile(hov, (N + 1, 1)); ubar = np.full((N, 4), o.HOV_W)
        qp = o.build_qp(xbar, ubar, x0, yref, yref_e, jac=o.jac_fd)
        H, h, _, _ = o.condense(qp)
        yield H, h, qp.lb.reshape(-1), qp.ub.reshape(-1), 0

which = sys.argv[1]
variants = [('current', {}), ('clip_absorb', dict(m=0.05)), ('clip_absorb', dict(m=0.01)), ('clip_absorb_mu', dict(m=0.05)), ('clip_mu', dict(m=0.05)), ('clip_mu', dict(m=0.02))]
def qps_st2():
    d = np.load('gpurun_out/st2_cases.npz')
    for i in range(len(d['it'])):
        qp = o.build_qp(d['xit'][i], d['uit'][i], d['x0'][i], yref, yref_e, jac=o.jac_fd)
        H, h, _, _ = o.condense(qp)
        yield H, h, qp.lb.reshape(-1), qp.ub.reshape(-1), int(d['it'][i])
qs = list(qps_st2()) if which == 'st2' else list(qps_hard()) if which == 'hard' else list(qps_typ(int(sys.argv[2]), float(sys.argv[3])))
ref = None
for name, kw in variants:
    its = []
    for H, h, lb, ub, gi in qs:
        try:
            it, v = ipm(H, h, lb, ub, name, **kw)
        except np.linalg.LinAlgError:
            it = -999
        if it != 0: its.append(it)
    its = np.array(its)
    print(f"{name:16s} {kw}: n {len(its)} mean {np.abs(its).mean():.2f} median {np.median(np.abs(its)):.0f} max {np.abs(its).max()} failed {(its < 0).sum()}")

print("---- violation of the unconstrained minimiser and iterations per start (current | clip_absorb_mu m=0.05)")
rows = []
for H, h, lb, ub, gi in qs:
    v0 = np.linalg.solve(H, -h)
    viol = max(np.maximum(lb - v0, 0).max(), np.maximum(v0 - ub, 0).max())
    if viol <= 0: continue
    try:
        a, _ = ipm(H, h, lb, ub, 'current'); b, _ = ipm(H, h, lb, ub, 'clip_absorb_mu', m=0.05)
    except np.linalg.LinAlgError:
        continue
    rows.append((viol, abs(a), abs(b)))
rows = np.array(rows)
for lo, hi in ((0, 2), (2, 5), (5, 11), (11, 22), (22, 44), (44, 100), (100, 1e9)):
    m = (rows[:, 0] >= lo) & (rows[:, 0] < hi)
    if m.any(): print(f"viol [{lo}, {hi}): n {m.sum()}  current {rows[m, 1].mean():.1f}  clipped {rows[m, 2].mean():.1f}")
