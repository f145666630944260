"""Round-3 GPU tool: captures QPs (iterate, x0) whose interior-point fall-back ended at the iteration cap (status 2) from the
engine's closed loop at twice the bench's disturbances -> gpurun_out/st2_cases.npz (source of tests/golden/hard_qps.npz)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda', 0)
f = bench.Fleet(32768, dev, np.random.default_rng(3), 'hover', 2.0)
for t in range(30):
    f.step()
torch.cuda.synchronize()
out = []
for t in range(8):
    xit, uit = f.solver.get_iterate()
    x0 = f.x.cpu().numpy().copy()
    # the step kicks a cohort before solving: replicate to know x0 actually used
    f.step(); torch.cuda.synchronize()
    st, it, rs = f.solver.stats()
    xn, un = f.solver.get_iterate()
    x0u = xn[:, 0]            # new iterate stage 0 = x0 used (for status 0); for failed rows the iterate is kept...
    sel = np.nonzero(st == 2)[0]
    print(t, 'status2', len(sel), 'status4', int((st == 4).sum()), 'iters of st2', it[sel][:10], 'res', rs[sel][:5])
    # x0 of this step = plant state before the step = f.xn (swapped) -> use solver's own x0 via get_x? keep plant state copy
    for i in sel[:8]:
        out.append((i, xit[i], uit[i], it[i], rs[i]))
    xprev = f.xn.cpu().numpy()   # after swap, xn holds the state the step started from (kicked)
    for k, i in enumerate(sel[:8]):
        out[-len(sel[:8]) + k] = out[-len(sel[:8]) + k] + (xprev[i].copy(),)
np.savez('gpurun_out/st2_cases.npz', idx=np.array([o[0] for o in out]), xit=np.array([o[1] for o in out]), uit=np.array([o[2] for o in out]),
         it=np.array([o[3] for o in out]), res=np.array([o[4] for o in out]), x0=np.array([o[5] for o in out]))
print('saved', len(out))
