"""Development aid: isolated timing of the interior-point sweeps (factor / forward / resolve) with
one or two waves per SIMD, for a library built with -DCFN_PROF (the timing variants quoted in DESIGN.md
-- factor stage without the LDS transpose / stores / loads / 4x4 inverse -- were one-off edits of
factor_stage, git history of this file's commit).
    python tools/sweep_bench.py lib0.so [lib1.so ...]"""
import os, sys, ctypes as C, subprocess
if len(sys.argv) > 2:   # one process per library (the library is loaded once per process)
    for l in sys.argv[1:]:
        subprocess.run([sys.executable, __file__, l])
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import crazyflie_nmpc_amd._lib as _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from crazyflie_nmpc_amd import BatchSolver, default_opts
from crazyflie_nmpc_amd.solver import INIT_HOVER
from crazyflie_nmpc_amd.synthetic import regulation_row, sample_hover_x0
L = _lib.lib()
L.cfnmpc_debug_bench_sweep.restype = C.c_float
L.cfnmpc_debug_bench_sweep.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
B, N = 4096 * 4, 50
rng = np.random.default_rng(1)
x = sample_hover_x0(rng, B)
row = regulation_row()
s = BatchSolver(B)
s.set_x0(x); s.set_yref(np.tile(row, (B, N, 1)), np.tile(row[:13], (B, 1))); s.init_iterate(INIT_HOVER)
s.solve(1)   # fills A, B, K, Rh, g ... with sane numbers
torch.cuda.synchronize()
out = [os.path.basename(sys.argv[1])]
for waves in (1024, 2048):
    for which, name in ((0, "factor"), (1, "forward"), (2, "resolve"), (3, "factor_as"), (4, "forward_as")):
        for head in (16, 50):
            reps = 20
            ms = L.cfnmpc_debug_bench_sweep(s._h, waves, head, reps, which)
            out.append(f"{waves // 1024}w/SIMD {name} head {head}: {ms * 1e3 / (reps * head):.3f} us/stage")
print("\n   ".join(out))
