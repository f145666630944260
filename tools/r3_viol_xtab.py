"""Round-3 GPU diagnostic behind cfnmpc_opts.as_skip_viol: violation of the unconstrained minimiser against the outcome of
the active-set iteration (settled / interior-point fall-back / failed), closed loop at 2x and 3x the bench's disturbances."""
import os, sys, ctypes as C
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))) if os.path.exists('/root/repo/bench.py') else '.')
import numpy as np, torch, bench
dev = torch.device('cuda', 0)
for sc in (2.0, 3.0):
    f = bench.Fleet(65536, dev, np.random.default_rng(3), 'hover', sc)
    for t in range(30): f.step()
    torch.cuda.synchronize()
    edges = [0, 11, 22, 44, 88, 176, 1e9]
    tab = np.zeros((len(edges) - 1, 4))
    for t in range(10):
        f.step(); torch.cuda.synchronize()
        st, it, _ = f.solver.stats()
        v = np.empty(65536); assert f.solver._L.cfnmpc_debug_get_viol(f.solver._h, v.ctypes.data_as(C.c_void_p)) == 0
        for b in range(len(edges) - 1):
            m = (v > edges[b]) & (v <= edges[b + 1])
            tab[b] += [m.sum(), (m & (it <= 12) & (st == 0)).sum(), (m & (it > 12) & (st == 0)).sum(), (m & (st != 0)).sum()]
    print(f"kick {sc}: viol bin | instances/step | AS settled | IPM converged | status != 0")
    for b in range(len(edges) - 1):
        print(f"   ({edges[b]}, {edges[b+1]}]: {tab[b,0]/10:8.1f} {tab[b,1]/10:8.1f} {tab[b,2]/10:8.1f} {tab[b,3]/10:8.1f}   mean AS solves n/a")
    f.close(); del f; torch.cuda.empty_cache()
