"""Round-3 numpy experiment: the active-set iteration on captured fall-back QPs neither settles nor cycles within 80 solves."""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from oracle import cfnmpc_oracle as o
d = np.load('gpurun_out/hard_cases.npz')
N = 50
yref, yref_e = o.regulation_yref(N, (0, 0, 0.4))
def pdas(H, h, lb, ub, maxs=80):
    v0 = np.linalg.solve(H, -h)
    lo, up = v0 < lb, v0 > ub
    seen = {}
    for s in range(1, maxs + 1):
        act = lo | up; free = ~act
        v = np.where(lo, lb, np.where(up, ub, 0.0))
        v[free] = np.linalg.solve(H[np.ix_(free, free)], -h[free] - H[np.ix_(free, act)] @ v[act])
        grad = H @ v + h
        lo2 = (free & (v < lb)) | (lo & (grad > 0)); up2 = (free & (v > ub)) | (up & (grad < 0))
        if np.array_equal(lo2, lo) and np.array_equal(up2, up):
            return s, v, None
        key = (lo2.tobytes(), up2.tobytes())
        if key in seen:
            return -s, v, s - seen[key]
        seen[key] = s
        lo, up = lo2, up2
    return 0, v, None
for i in range(len(d['it'])):
    qp = o.build_qp(d['xit'][i], d['uit'][i], d['x0'][i], yref, yref_e, jac=o.jac_fd)
    H, h, Gam, g = o.condense(qp)
    lb, ub = qp.lb.reshape(-1), qp.ub.reshape(-1)
    s, v, cyc = pdas(H, h, lb, ub)
    ev = np.linalg.eigvalsh(H)
    print(i, 'gpu ipm iters', d['it'][i], 'pdas', s, 'cycle len', cyc, 'cond', f'{ev[-1]/ev[0]:.2e}', 'nact', int(((v<=lb+1e-9)|(v>=ub-1e-9)).sum()))
