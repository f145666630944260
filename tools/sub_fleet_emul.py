"""Development experiment: ONE fused RTI step of the fleet as n sub-fleets -- the start solves (k_linfactor, VALU-bound,
memory-light) back to back on one stream, each sub-fleet's forward sweep + constrained QPs (memory / latency-bound) on its
own stream behind its start solve; fork / join per step as a library call would do it.
    python tools/sub_fleet_emul.py [batch]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch
import cfnmpc_oracle as o
from crazyflie_nmpc_amd import BatchSolver, default_opts, sim
from crazyflie_nmpc_amd.solver import INIT_HOVER

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda", 0)
def make(Bh, seed, mode):
    rng = np.random.default_rng(seed)
    x0 = torch.from_numpy(o.sample_hover_x0(rng, Bh)).to(dev)
    yr, ye = o.regulation_yref(50, (0, 0, 0.4))
    s = BatchSolver(Bh, default_opts(start_solve=mode, as_passes=-1, forward_sweep=1))
    s.set_yref(np.repeat(yr[None], Bh, 0).copy(), np.repeat(ye[None], Bh, 0).copy())
    s.set_x0(x0); s.init_iterate(INIT_HOVER)
    return dict(s=s, x=x0, xn=torch.empty_like(x0), u=torch.empty((Bh, 4), dtype=torch.float64, device=dev))
def plant(f, t):
    f["s"].get_u(0, f["u"])
    sim(f["x"], f["u"], T=0.015, steps=1, out=f["xn"])
    f["x"], f["xn"] = f["xn"], f["x"]
    if t % 10 == 9:
        f["x"][:, 7:10] += 0.3 * torch.randn((f["x"].shape[0], 3), dtype=torch.float64, device=dev)
def bench(label, stepfn, fl, n=30):
    for t in range(10): stepfn(t)
    torch.cuda.synchronize()
    t0 = time.time()
    for t in range(n): stepfn(t)
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    print(f"{label}: {dt * 1e3:.3f} ms per step = {B / dt / 1e6:.2f} M steps/s", flush=True)
    for f in fl: f["s"].close()

for mode in (1, 2):
    fl = [make(B, 7, mode)]
    def step(t):
        f = fl[0]; f["s"].set_x0(f["x"]); f["s"].solve(1); plant(f, t)
    bench(f"start_solve {mode}, whole fleet, one stream", step, fl)

main = torch.cuda.current_stream(dev)
for nsub in (2, 4, 8):
    for half in (0, 1):
        fl = [make(B // nsub, 7 + i, 2) for i in range(nsub)]
        lf = torch.cuda.Stream(dev); qs = [torch.cuda.Stream(dev) for _ in range(nsub)]
        L = fl[0]["s"]._L
        def step(t):
            for f in fl: f["s"].set_x0(f["x"])
            fork = torch.cuda.Event(); fork.record(main)
            lf.wait_event(fork)
            for f, q in zip(fl, qs):
                assert L.cfnmpc_debug_solve_part(f["s"]._h, 1, 0, C.c_void_p(lf.cuda_stream)) == 0
                e = torch.cuda.Event(); e.record(lf)
                q.wait_event(e)
                assert L.cfnmpc_debug_solve_part(f["s"]._h, 2, half, C.c_void_p(q.cuda_stream)) == 0
                d = torch.cuda.Event(); d.record(q); main.wait_event(d)
            for f in fl: plant(f, t)
        bench(f"fused, {nsub} sub-fleets, fork/join per step, forward at {'2 waves' if half else '1 wave'}/SIMD", step, fl)
