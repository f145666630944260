"""Development experiment: the fleet as TWO half-fleets on two streams, out of phase (the fused start solve is VALU-bound
and memory-light, the forward sweep / active-set kernels memory-bound: do they overlap?).
    python tools/two_stream.py <start_solve> [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch
import cfnmpc_oracle as o
from crazyflie_nmpc_amd import BatchSolver, default_opts, sim
from crazyflie_nmpc_amd.solver import INIT_HOVER

mode = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device("cuda", 0)
def make(Bh, seed):
    rng = np.random.default_rng(seed)
    x0 = torch.from_numpy(o.sample_hover_x0(rng, Bh)).to(dev)
    yr, ye = o.regulation_yref(50, (0, 0, 0.4))
    s = BatchSolver(Bh, default_opts(start_solve=mode, as_passes=-1, forward_sweep=1))
    s.set_yref(np.repeat(yr[None], Bh, 0).copy(), np.repeat(ye[None], Bh, 0).copy())
    s.set_x0(x0); s.init_iterate(INIT_HOVER)
    return dict(s=s, x=x0, xn=torch.empty_like(x0), u=torch.empty((Bh, 4), dtype=torch.float64, device=dev), rng=rng)
def step(f, t):
    f["s"].set_x0(f["x"]); f["s"].solve(1); f["s"].get_u(0, f["u"])
    sim(f["x"], f["u"], T=0.015, steps=1, out=f["xn"])
    f["x"], f["xn"] = f["xn"], f["x"]
    if t % 10 == 9:   # the bench's disturbance, roughly
        f["x"][:, 7:10] += 0.3 * torch.randn((f["x"].shape[0], 3), dtype=torch.float64, device=dev)
for nsplit in (1, 2, 3, 4):
    per = (B // nsplit) // 64 * 64
    fl = [make(per if i < nsplit - 1 else B - per * (nsplit - 1), 7 + i) for i in range(nsplit)]
    st = [torch.cuda.Stream(dev) for _ in range(nsplit)]
    def run(n):
        for t in range(n):
            for f, s_ in zip(fl, st):
                with torch.cuda.stream(s_):
                    step(f, t)
    run(10); torch.cuda.synchronize()
    t0 = time.time(); run(30); torch.cuda.synchronize(); dt = (time.time() - t0) / 30
    print(f"start_solve {mode}  {nsplit} stream(s) x ~{B // nsplit}: {dt * 1e3:.3f} ms per step of the whole fleet = {B / dt / 1e6:.2f} M steps/s", flush=True)
    for f in fl: f["s"].close()
