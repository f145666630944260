"""Closed-loop step time WITHOUT the per-kernel event profiling of bench.py (steps timed with cfnmpc_set_profiling are launched
individually): cfnmpc_opts.step_graph 0 / 1 at the strong-scaling shard sizes.  usage: python tools/step_graph_time.py [batches ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
dev = torch.device("cuda", 0)
for B in [int(a) for a in sys.argv[1:]] or [2048, 4096, 8192, 16384]:
    for rep in range(2):
        for g in (0, 1):
            f = bench.Fleet(B, dev, np.random.default_rng(3), "hover", 1.0, step_graph=g)
            for _ in range(40):
                f.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                f.step()
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / 100
            st = f.solver.stats()[0]
            print(f"batch {B} step_graph {g}: {el * 1e3:.4f} ms/step  {B / el / 1e6:.3f} M steps/s  ok {float((st == 0).mean()):.3f}")
            f.close(); del f; torch.cuda.empty_cache()
