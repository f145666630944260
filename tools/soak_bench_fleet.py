"""Soak run (development aid): a long closed loop of the bench fleet -- statuses, iterate sanity, device memory, step time
at the start and at the end.    python tools/soak_bench_fleet.py [steps] [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
wl = sys.argv[2] if len(sys.argv) > 2 else "hover"
dev = torch.device("cuda", 0)
f = bench.Fleet(65536, dev, np.random.default_rng(3), wl, 1.0)
free0 = torch.cuda.mem_get_info()[0]
bad = 0; tms = []
for blk in range(steps // 100):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(100): f.step()
    torch.cuda.synchronize(); tms.append((time.perf_counter() - t0) * 10)
    st, it, res = f.solver.stats()
    x, u = f.solver.get_iterate()
    nb = int((st != 0).sum()); bad += nb
    assert np.isfinite(x).all() and np.isfinite(u).all(), blk
    assert u.min() >= -1e-9 and u.max() <= 22.0 + 1e-9, (u.min(), u.max())
    if blk % 5 == 0:
        print(f"steps {blk * 100 + 100}: {tms[-1]:.3f} ms/step, status != 0: {nb}, |pos - target| median {np.median(np.abs(x[:, 0, :3] - x[:, -1, :3])):.3f}, free mem delta {(free0 - torch.cuda.mem_get_info()[0]) / 2**20:.1f} MiB", flush=True)
print(f"{wl}: {steps} steps, failed row-steps {bad}, ms/step first / last block {tms[0]:.3f} / {tms[-1]:.3f}, min {min(tms):.3f} max {max(tms):.3f}")
f.close()
