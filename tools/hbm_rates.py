"""Fill / copy / read rates of the box (GB/s) next to the bench's kernel times: tells a write-limited box from the others
(profiles/r05_linearise_stores.md).  python tools/hbm_rates.py [GiB]"""
import sys, torch
g = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
n = int(g * (1 << 30) // 8)
a = torch.empty(n, dtype=torch.float64, device="cuda"); b = torch.empty_like(a)
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
tf = t(lambda: a.fill_(1.0)); tc = t(lambda: b.copy_(a)); tr = t(lambda: a.sum())
print("hbm_rates GiB=%.1f fill %.0f GB/s  copy %.0f GB/s (read+write)  read %.0f GB/s" % (g, n * 8 / tf / 1e9, 2 * n * 8 / tc / 1e9, n * 8 / tr / 1e9))
