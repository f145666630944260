"""Constrained-QP diagnostics of the closed-loop bench workload on the GPU: per kick scale the
distribution of active-set solves / interior-point iterations, head classes and statuses.
usage: python tools/qp_diag.py [scales ...]   (default 1 2 3)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench

dev = torch.device("cuda", 0)
scales = [float(a) for a in sys.argv[1:]] or [1.0, 2.0, 3.0]
B = int(os.environ.get("QPD_BATCH", "65536"))
for sc in scales:
    f = bench.Fleet(B, dev, np.random.default_rng(3), "hover", sc)
    for t in range(30):
        f.step()
    torch.cuda.synchronize()
    agg_it = np.zeros(64, dtype=np.int64); agg_hd = np.zeros(128, dtype=np.int64); nst = np.zeros(8, dtype=np.int64)
    for t in range(10):
        f.step(); torch.cuda.synchronize()
        st, it, rs = f.solver.stats(); hd = f.solver.heads()
        agg_it += np.bincount(np.minimum(it, 63), minlength=64)
        agg_hd += np.bincount(hd[it > 0], minlength=128)[:128]
        nst += np.bincount(st, minlength=8)[:8]
    print(f"== kick scale {sc}: per step (mean of 10): constrained {(agg_it[1:].sum())/10:.0f} of {B}")
    print("   solves/iterations histogram:", {i: int(c / 10) for i, c in enumerate(agg_it) if c and i > 0})
    print("   iters > 12 (interior-point fall-back):", agg_it[13:].sum() / 10, " status counts", (nst / 10).tolist())
    print("   head classes of constrained:", {i: int(c / 10) for i, c in enumerate(agg_hd) if c})
    f.close(); del f; torch.cuda.empty_cache()
