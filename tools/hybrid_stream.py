"""Development experiment: the fleet as two free-running sub-fleets on two streams, one on the FUSED start solve (VALU-bound, few bytes),
one on the stored-block path (bandwidth-bound): do the two resource profiles overlap?   python tools/hybrid_stream.py [fused share ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch
import cfnmpc_oracle as o
from crazyflie_nmpc_amd import BatchSolver, default_opts, sim
from crazyflie_nmpc_amd.solver import INIT_HOVER
B = 65536
dev = torch.device("cuda", 0)
def make(Bh, seed, mode):
    rng = np.random.default_rng(seed)
    x0 = torch.from_numpy(o.sample_hover_x0(rng, Bh)).to(dev)
    yr, ye = o.regulation_yref(50, (0, 0, 0.4))
    s = BatchSolver(Bh, default_opts(start_solve=mode, as_passes=-1, forward_sweep=1))
    s.set_yref(np.repeat(yr[None], Bh, 0).copy(), np.repeat(ye[None], Bh, 0).copy())
    s.set_x0(x0); s.init_iterate(INIT_HOVER)
    return dict(s=s, x=x0, xn=torch.empty_like(x0), u=torch.empty((Bh, 4), dtype=torch.float64, device=dev))
def step(f, t):
    f["s"].set_x0(f["x"]); f["s"].solve(1); f["s"].get_u(0, f["u"])
    sim(f["x"], f["u"], T=0.015, steps=1, out=f["xn"])
    f["x"], f["xn"] = f["xn"], f["x"]
    if t % 10 == 9:
        f["x"][:, 7:10] += 0.3 * torch.randn((f["x"].shape[0], 3), dtype=torch.float64, device=dev)
def run(fl, st, n):
    for t in range(n):
        for f, s_ in zip(fl, st):
            with torch.cuda.stream(s_):
                step(f, t)
for share in [float(a) for a in (sys.argv[1:] or ["0", "0.25", "0.4", "0.5", "1"])]:
    nf = int(B * share) // 64 * 64
    fl = ([make(nf, 7, 2)] if nf else []) + ([make(B - nf, 8, 1)] if B - nf else [])
    st = [torch.cuda.Stream(dev) for _ in fl]
    run(fl, st, 10); torch.cuda.synchronize()
    t0 = time.time(); run(fl, st, 30); torch.cuda.synchronize(); dt = (time.time() - t0) / 30
    print(f"fused share {share:.2f} ({nf} fused + {B - nf} stored): {dt * 1e3:.3f} ms per step = {B / dt / 1e6:.2f} M steps/s", flush=True)
    for f in fl: f["s"].close()
