"""Clipped start of the interior point (cfnmpc_opts.ipm_clip_viol) on QPs captured from tumbling vehicles
(tests/golden/hard_qps.npz: unconstrained minimiser 100 - 6000 kRPM outside the box, condition 1e9 - 1e12): the HIP
interior point follows the numpy oracle and the C restatement iteration count by iteration count, converges where
the infeasible start hits the cap, and ordinary QPs keep the infeasible start."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def _objective(oracle, qp, du):
    H, h, _, _ = oracle.condense(qp)
    v = du.reshape(-1)
    return 0.5 * v @ H @ v + h @ v


def test_clipped_start_matches_oracles_on_captured_qps(oracle, cref):
    from crazyflie_nmpc_amd import BatchSolver, default_opts
    q = np.load(os.path.join(G, "hard_qps.npz"))
    n, N = q["x0"].shape[0], 50
    yr, ye = oracle.regulation_yref(N, (0.0, 0.0, 0.4))
    yref = np.repeat(yr[None], n, 0).copy(); yref_e = np.repeat(ye[None], n, 0).copy()
    res = {}
    for clip in (2.0, 0.0):
        s = BatchSolver(n, default_opts(active_set=0, active_horizon=0, max_iter=100, ipm_clip_viol=clip))
        s.set_yref(yref, yref_e); s.set_iterate(q["xit"], q["uit"]); s.set_x0(q["x0"])
        s.solve(1)
        st, it, _ = s.stats()
        xg, ug = s.get_iterate()
        res[clip] = (st.copy(), it.copy(), ug - q["uit"])
    st, it, du = res[2.0]
    assert (st == 0).all()
    assert np.abs(it - q["iters_clipped_start"]).max() <= 1, (it, q["iters_clipped_start"])     # the numpy oracle's counts
    assert it.max() <= 30
    # the infeasible start needs 26 - 100 iterations on the same QPs (about the numpy oracle's counts: long runs of tiny
    # steps at condition 1e11 drift by a few iterations between implementations)
    st0, it0, _ = res[0.0]
    ref0 = np.minimum(np.abs(q["iters_infeasible_start"]), 100)
    assert (np.abs(it0 - ref0) <= np.maximum(3, 0.15 * ref0)).all(), (it0, ref0)
    assert it0.mean() > 2 * it.mean()
    # same minimum: objective of the condensed QP (the argmin itself is determined to ~1e-2 at condition 1e11)
    for i in range(n):
        qp = oracle.build_qp(q["xit"][i], q["uit"][i], q["x0"][i], yr, ye)
        f = _objective(oracle, qp, du[i])
        assert abs(f - q["objective"][i]) <= 1e-7 * abs(q["objective"][i]), (i, f, q["objective"][i])
        assert (du[i] >= qp.lb - 1e-7).all() and (du[i] <= qp.ub + 1e-7).all()
    # C restatement: the same iteration, count by count
    opts = cref.default_opts(max_iter=100)
    xr, ur = q["xit"].copy(), q["uit"].copy()
    st_r, it_r, _, _ = cref.rti_step(opts, xr, ur, q["x0"].copy(), yref, yref_e, nthreads=0)
    assert (st_r == 0).all() and np.abs(it - it_r).max() <= 1, (it, it_r)


def test_fallback_after_active_set_uses_the_clipped_start(oracle):
    """Default options: the active-set iteration does not settle on these QPs (12 solves), the interior-point fall-back
    starts clipped and converges well below the cap of 50 -- with the infeasible start most of them end at the cap."""
    from crazyflie_nmpc_amd import BatchSolver, default_opts
    q = np.load(os.path.join(G, "hard_qps.npz"))
    n, N = q["x0"].shape[0], 50
    yr, ye = oracle.regulation_yref(N, (0.0, 0.0, 0.4))
    out = {}
    for clip in (2.0, 0.0):
        s = BatchSolver(n, default_opts(ipm_clip_viol=clip))
        s.set_yref(np.repeat(yr[None], n, 0).copy(), np.repeat(ye[None], n, 0).copy()); s.set_iterate(q["xit"], q["uit"]); s.set_x0(q["x0"])
        s.solve(1)
        st, it, _ = s.stats()
        out[clip] = (st.copy(), it.copy())
    st, it = out[2.0]
    assert (st == 0).all() and it.max() <= 32 and it.min() >= 13        # interior-point iterations (more than 12 solves)
    st0, it0 = out[0.0]
    assert (st0 != 0).sum() >= 5 and it0.mean() > it.mean() + 10


def test_ordinary_qps_keep_the_infeasible_start(oracle):
    """Below the threshold nothing changes: bitwise the results of ipm_clip_viol = 0."""
    from crazyflie_nmpc_amd import BatchSolver, default_opts
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    B, N = 300, 50
    x0 = oracle.sample_hover_x0(np.random.default_rng(3), B, scale=2.0)
    yr, ye = oracle.regulation_yref(N, (0.0, 0.0, 0.4))
    r = []
    for clip in (2.0, 0.0):
        s = BatchSolver(B, default_opts(active_set=0, ipm_clip_viol=clip))
        s.set_yref(np.repeat(yr[None], B, 0).copy(), np.repeat(ye[None], B, 0).copy()); s.set_x0(x0); s.init_iterate(INIT_HOVER)
        s.solve(1)
        r.append((s.get_iterate(), s.stats()))
    assert np.array_equal(r[0][0][1], r[1][0][1]) and np.array_equal(r[0][1][1], r[1][1][1])
    assert (r[0][1][1] > 0).sum() > 100
    with pytest.raises(Exception):
        BatchSolver(4, default_opts(ipm_clip_margin=0.6))
