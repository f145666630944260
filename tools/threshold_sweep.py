"""Cross-overs of the kernel choices cfnmpc_create makes by fleet size, measured per horizon (profiles/r04_thresholds.md):
closed-loop ms per RTI step for forward_sweep in {1 matrix-free, 2 row groups} x as_passes in {-1 monolithic, -3 solves + commit}.
    python tools/threshold_sweep.py [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch
import cfnmpc_oracle as o
from crazyflie_nmpc_amd import BatchSolver, default_opts, sim
from crazyflie_nmpc_amd.solver import INIT_HOVER

dev = torch.device("cuda", 0)
P = 20
def run(B, N, **kw):
    rng = np.random.default_rng(5)
    s = BatchSolver(B, default_opts(N=N, **kw))
    yr, ye = o.regulation_yref(N, (0, 0, 0.4))
    s.set_yref(np.repeat(yr[None], B, 0).copy(), np.repeat(ye[None], B, 0).copy())
    x = torch.from_numpy(o.sample_hover_x0(rng, B)).to(dev); xn = torch.empty_like(x)
    u0 = torch.empty((B, 4), dtype=torch.float64, device=dev)
    cohort = (B + P - 1) // P
    kicks = torch.from_numpy(o.sample_hover_x0(rng, cohort * P).reshape(P, cohort, 13)).to(dev)
    s.set_x0(x); s.init_iterate(INIT_HOVER)
    t = 0
    def step():
        nonlocal x, xn, t
        c0 = (t % P) * cohort; c1 = min(c0 + cohort, B)
        if c1 > c0: x[c0:c1].copy_(kicks[t % P, : c1 - c0])
        s.set_x0(x); s.solve(1); s.get_u(0, u0); sim(x, u0, T=0.015, steps=1, out=xn); x, xn = xn, x; t += 1
    for _ in range(20): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 40 * 1e3
    ok = float((s.stats()[0] == 0).mean())
    s.close()
    return ms, ok

Ns = [int(a) for a in sys.argv[1:]] or [30, 50, 100]
print("| N | batch | fs=1 ap=-1 | fs=1 ap=-3 | fs=2 ap=-1 | fs=2 ap=-3 | auto | best |")
print("|---|---|---|---|---|---|---|---|")
for N in Ns:
    for B in (2048, 4096, 6144, 8192, 12288, 16384, 24576, 32768, 49152, 65536):
        if B * N > 65536 * 60 and N == 100 and B > 49152: pass
        res = {}
        for fs in (1, 2):
            for ap in (-1, -3):
                res[(fs, ap)] = run(B, N, forward_sweep=fs, as_passes=ap)[0]
        auto = run(B, N)[0]
        best = min(res, key=res.get)
        print(f"| {N} | {B} | " + " | ".join(f"{res[k]:.4f}" for k in ((1, -1), (1, -3), (2, -1), (2, -3))) + f" | {auto:.4f} | fs={best[0]} ap={best[1]} |", flush=True)
