# closed-loop throughput under heavy disturbances (development aid; pass cfnmpc_opts overrides in OPTS, e.g. OPTS="as_skip_viol=2.0")
cd $GRAFT_REPO_ROOT
python - <<'PY'
import numpy as np, torch, time, sys, os
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda", 0)
for kick in (1.0, 2.0, 3.0):
    for rep in (0, 1):
        kw = {k: float(v) if '.' in v else int(v) for k, v in (a.split('=') for a in os.environ.get('OPTS', '').split())}
        f = bench.Fleet(65536, dev, np.random.default_rng(3), "hover", kick, **kw)
        for t in range(15): f.step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(20): f.step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        st = f.solver.stats()[0]
        print(f"kick x{kick} {kw}: {dt * 1e3:.3f} ms/step  {65536 / dt / 1e6:.3f} M steps/s  ok {float((st == 0).mean()):.5f}", flush=True)
        f.close()
PY
