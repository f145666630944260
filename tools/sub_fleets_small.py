"""Development experiment: a SMALL fleet (strong-scaling shard sizes) as n sub-fleets on n streams, free-running -- where the
step's kernels are latency chains with idle SIMDs, do independent sub-fleets fill them?
    python tools/sub_fleets_small.py [batch ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch
import cfnmpc_oracle as o
from crazyflie_nmpc_amd import BatchSolver, default_opts, sim
from crazyflie_nmpc_amd.solver import INIT_HOVER

dev = torch.device("cuda", 0)
def make(Bh, seed):
    rng = np.random.default_rng(seed)
    x0 = torch.from_numpy(o.sample_hover_x0(rng, Bh)).to(dev)
    yr, ye = o.regulation_yref(50, (0, 0, 0.4))
    s = BatchSolver(Bh, default_opts())
    s.set_yref(np.repeat(yr[None], Bh, 0).copy(), np.repeat(ye[None], Bh, 0).copy())
    s.set_x0(x0); s.init_iterate(INIT_HOVER)
    return dict(s=s, x=x0, xn=torch.empty_like(x0), u=torch.empty((Bh, 4), dtype=torch.float64, device=dev))
def step(f, t):
    f["s"].set_x0(f["x"]); f["s"].solve(1); f["s"].get_u(0, f["u"])
    sim(f["x"], f["u"], T=0.015, steps=1, out=f["xn"])
    f["x"], f["xn"] = f["xn"], f["x"]
    if t % 10 == 9:   # the bench's disturbance, roughly
        f["x"][:, 7:10] += 0.3 * torch.randn((f["x"].shape[0], 3), dtype=torch.float64, device=dev)
for B in [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384]:
    for nsplit in [int(a) for a in os.environ.get("NSPLIT", "1,2,4,1,2,4").split(",")]:
        per = B // nsplit
        fl = [make(per, 7 + i) for i in range(nsplit)]
        st = [torch.cuda.Stream(dev) for _ in range(nsplit)]
        def run(n):
            for t in range(n):
                for f, s_ in zip(fl, st):
                    with torch.cuda.stream(s_):
                        step(f, t)
        run(20); torch.cuda.synchronize()
        t0 = time.time(); run(60); torch.cuda.synchronize(); dt = (time.time() - t0) / 60
        # host-only rate of the same loop (how far the GPU is from being the bound)
        print(f"{B} instances as {nsplit} x {per}: {dt * 1e3:.4f} ms per step = {B / dt / 1e6:.2f} M steps/s", flush=True)
        for f in fl: f["s"].close()
