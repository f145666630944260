"""Head-condensed dense active-set solves (cfnmpc_opts.as_dense, csrc/cfnmpc_asdense.hip) against the Riccati form of the same
iteration (as_dense = -1: k_as_solves for every row) and against the CPU restatement: same statuses, same solve counts, same
iterates to rounding -- closed loops with staggered kicks (heads of 4 .. 16 stages and longer ones side by side), horizons
around and below the dense limit (the head is the whole horizon and ends in the terminal weight), other boxes and weights."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HOV = 15.777730167256925


def _loop(oracle, B, N, steps, scale, opt_a, opt_b, seed=5, box=None, check=None):
    from crazyflie_nmpc_amd import BatchSolver, default_opts, sim
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    rng = np.random.default_rng(seed)
    x = oracle.sample_hover_x0(rng, B, scale=scale)
    yr, ye = oracle.regulation_yref(N, (0.0, 0.0, 0.4))
    yref = np.repeat(yr[None], B, 0).copy(); yref_e = np.repeat(ye[None], B, 0).copy()
    a = BatchSolver(B, default_opts(N=N, **opt_a)); b = BatchSolver(B, default_opts(N=N, **opt_b))
    for s in (a, b):
        if box:
            s.set_box(*box)
        s.set_x0(x); s.set_yref(yref, yref_e); s.init_iterate(INIT_HOVER)
    cohort = max(1, B // 10)
    kicks = oracle.sample_hover_x0(rng, cohort * steps, scale=scale).reshape(steps, cohort, 13)
    heads, total = {}, 0
    for t in range(steps):
        c0 = (t * cohort) % max(B - cohort, 1)
        x[c0:c0 + cohort] = kicks[t]
        a.set_x0(x); b.set_x0(x)
        a.solve(1); b.solve(1)
        sa, ia, _ = a.stats(); sb, ib, _ = b.stats()
        xa, ua = a.get_iterate(); xb, ub = b.get_iterate()
        assert np.array_equal(sa, sb), (t, np.nonzero(sa != sb)[0][:10], sa[sa != sb][:10], sb[sa != sb][:10])
        assert np.array_equal(ia, ib), (t, np.nonzero(ia != ib)[0][:10], ia[ia != ib][:10], ib[ia != ib][:10])
        ok = sa == 0
        err = max(np.abs(ua[ok] - ub[ok]).max(), np.abs(xa[ok] - xb[ok]).max())
        assert err < 1e-8, (t, err)
        for h in a.heads()[ia > 0]:
            heads[int(h)] = heads.get(int(h), 0) + 1
        total += int((ia > 0).sum())
        if check:
            check(t, x, a, ia, sa, ua, xa)
        b.set_iterate(xa, ua)                      # one trajectory
        x = sim(x, a.get_u(0), T=0.015, steps=1)
    a.close(); b.close()
    return heads, total


@pytest.mark.parametrize("B,scale", [(37, 1.0), (1500, 1.0), (1500, 1.6), (6000, 1.0)])
def test_dense_solves_match_riccati_solves(oracle, B, scale):
    heads, total = _loop(oracle, B, 50, 12, scale, dict(as_dense=1), dict(as_dense=-1, as_passes=-3))
    assert total > B // 4 and sum(v for h, v in heads.items() if h <= 16) > 0.5 * total, heads     # the dense kernel had the bulk of the rows
    if scale > 1.5:
        assert any(h > 16 for h in heads), heads                                                  # ... beside rows of k_as_solves


@pytest.mark.parametrize("N", [5, 8, 9, 12, 16, 17, 20])
def test_dense_solves_on_short_horizons(oracle, N):
    """N <= 16: every constrained row's head is the whole horizon (terminal weight instead of a checkpoint, 4 N inputs, padded to
    the next multiple of sixteen); N = 17, 20: heads 4 .. 16 dense, the full horizon with k_as_solves."""
    heads, total = _loop(oracle, 300, N, 6, 2.0, dict(as_dense=1), dict(as_dense=-1, as_passes=-3), seed=N)
    assert total > 50, (heads, total)


def test_dense_solves_match_restatement(oracle, cref):
    """... and against the C restatement (full-horizon active-set solves there): statuses, solve counts where both sides sweep the
    full horizon are covered by test_gpu_parity; here the default engine (active horizon, dense heads) closed loop, iterates 1e-7."""
    from crazyflie_nmpc_amd import BatchSolver, default_opts, sim
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    B, N = 256, 50
    rng = np.random.default_rng(77)
    x = oracle.sample_hover_x0(rng, B, scale=1.3)
    yr, ye = oracle.regulation_yref(N, (0.0, 0.0, 0.4))
    yref = np.repeat(yr[None], B, 0).copy(); yref_e = np.repeat(ye[None], B, 0).copy()
    # (tol 1e-11 on both sides: rows far outside the box skip the active-set iteration -- as_skip_viol -- and are solved by the
    #  interior point, whose two implementations agree at the level of the QP tolerance)
    s = BatchSolver(B, default_opts(as_dense=1, u_min=1.0, u_max=20.0, tol=1e-11))
    s.set_x0(x); s.set_yref(yref, yref_e); s.init_iterate(INIT_HOVER)
    opts = cref.default_opts(active_set=1, u_min=1.0, u_max=20.0, tol=1e-11)
    xr = np.repeat(x[:, None, :], N + 1, 1).copy(); ur = np.full((B, N, 4), HOV)
    n = 0
    for t in range(10):
        s.set_x0(x); s.solve(1)
        st, it, rs = s.stats()
        st_r, it_r, rs_r, _ = cref.rti_step(opts, xr, ur, x.copy(), yref, yref_e, nthreads=0)
        xg, ug = s.get_iterate()
        assert (st == 0).all() and (st_r == 0).all() and np.array_equal(it > 0, it_r > 0)
        exact = (rs == 0.0) & (rs_r == 0.0)     # settled by active-set solves on both sides (residual exactly 0; the interior point reports its own > 0)
        assert exact.mean() > 0.98
        assert np.abs(ug[exact] - ur[exact]).max() < 1e-7 and np.abs(xg[exact] - xr[exact]).max() < 1e-7, (t, np.abs(ug[exact] - ur[exact]).max())
        assert np.abs(ug - ur).max() < 5e-4
        assert ug.min() >= 1.0 - 1e-9 and ug.max() <= 20.0 + 1e-9
        n += int((it > 0).sum())
        ur[:] = ug; xr[:] = xg
        x = sim(x, s.get_u(0), T=0.015, steps=1)
    assert n > 100
    s.close()


@pytest.mark.parametrize("B,scale,N", [(600, 1.0, 50), (9000, 1.0, 50), (9000, 1.8, 50), (2000, 1.5, 100), (700, 2.5, 40)])
def test_split_forward_sweep_matches_single_launch(oracle, B, scale, N):
    """cfnmpc_opts.forward_split: the start solve's forward sweep in two launches (stages [0, 24) + classification | stages [24, N)
    beside the constrained rows' kernels, late rows appended to the list) against the single launch: statuses equal, the same
    instances constrained, iterates to rounding -- closed loops whose kicks make rows with heads of 24 / 32 / N stages and rows that
    first leave the box behind stage 24 (late rows)."""
    from crazyflie_nmpc_amd import BatchSolver, default_opts, sim
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    rng = np.random.default_rng(B + N)
    x = oracle.sample_hover_x0(rng, B, scale=scale)
    yr, ye = oracle.regulation_yref(N, (0.0, 0.0, 0.4))
    yref = np.repeat(yr[None], B, 0).copy(); yref_e = np.repeat(ye[None], B, 0).copy()
    a = BatchSolver(B, default_opts(N=N, as_dense=1, forward_sweep=1, forward_split=1))
    b = BatchSolver(B, default_opts(N=N, as_dense=1, forward_sweep=1, forward_split=-1))
    for s in (a, b):
        s.set_x0(x); s.set_yref(yref, yref_e); s.init_iterate(INIT_HOVER)
    cohort = max(1, B // 10)
    steps = 10
    kicks = oracle.sample_hover_x0(rng, cohort * steps, scale=scale).reshape(steps, cohort, 13)
    total = long_heads = 0
    for t in range(steps):
        c0 = (t * cohort) % max(B - cohort, 1)
        x[c0:c0 + cohort] = kicks[t]
        a.set_x0(x); b.set_x0(x)
        a.solve(1); b.solve(1)
        sa, ia, ra = a.stats(); sb, ib, rb = b.stats()
        xa, ua = a.get_iterate(); xb, ub = b.get_iterate()
        assert np.array_equal(sa, sb), (t, np.nonzero(sa != sb)[0][:10])
        assert np.array_equal(ia > 0, ib > 0), (t, np.nonzero((ia > 0) != (ib > 0))[0][:10])
        ok = sa == 0
        exact = ok & (ra == 0.0) & (rb == 0.0)     # active-set solves on both sides (an interior-point row reports its residual > 0)
        # (a row's head comes from the first window only, so a few rows take another path -- a retry over a longer head, or the
        #  interior point -- to the same solution: exact solves agree to rounding, interior-point rows to its tolerance)
        assert np.abs(ua[exact] - ub[exact]).max() < 1e-7 and np.abs(xa[exact] - xb[exact]).max() < 1e-7, (t, np.abs(ua[exact] - ub[exact]).max())
        assert np.abs(ua[ok] - ub[ok]).max() < 5e-4
        total += int((ia > 0).sum()); long_heads += int((a.heads()[ia > 0] > 24).sum())
        b.set_iterate(xa, ua)
        x = sim(x, a.get_u(0), T=0.015, steps=1)
    assert total > B // 5
    if scale >= 1.5:
        assert long_heads > 0
    a.close(); b.close()


@pytest.mark.parametrize("B", [9000, 17000])
def test_late_rows_of_the_split_sweep_reach_the_interior_point(oracle, B):
    """A LATE row of the split forward sweep (feasible over [0, 24), first violation behind it: appended to the list by part two)
    that the retry kernel does not settle must reach the interior-point fall-back -- round 5's launch loops stopped at the list's
    original length and left such a row with status 0 and an iterate outside the box (advisor).  Forced here with a tiny
    as_skip_viol: every constrained row skips the active-set iteration, so every late row depends on the fall-back.  9000:
    the fall-back loops over the list itself; 17 000 (>= 16 S instances): over the compacted fall-back list of k_ipm_list."""
    from crazyflie_nmpc_amd import BatchSolver, default_opts, sim
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    N = 50
    rng = np.random.default_rng(B)
    x = oracle.sample_hover_x0(rng, B, scale=1.8)
    yr, ye = oracle.regulation_yref(N, (0.0, 0.0, 0.4))
    yref = np.repeat(yr[None], B, 0).copy(); yref_e = np.repeat(ye[None], B, 0).copy()
    kw = dict(N=N, as_dense=1, forward_sweep=1, as_skip_viol=1e-6, tol=1e-11)
    a = BatchSolver(B, default_opts(forward_split=1, **kw)); b = BatchSolver(B, default_opts(forward_split=-1, **kw))
    for s in (a, b):
        s.set_x0(x); s.set_yref(yref, yref_e); s.init_iterate(INIT_HOVER)
    cohort = B // 10
    kicks = oracle.sample_hover_x0(rng, cohort * 6, scale=1.8).reshape(6, cohort, 13)
    late = 0
    for t in range(6):
        x[t * cohort:(t + 1) * cohort] = kicks[t]
        a.set_x0(x); b.set_x0(x)
        a.solve(1); b.solve(1)
        sa, ia, _ = a.stats(); sb, ib, _ = b.stats()
        xa, ua = a.get_iterate(); xb, ub = b.get_iterate()
        late += a.list_counts()[3]
        assert b.list_counts()[3] == 0
        assert np.array_equal(ia > 0, ib > 0), (t, np.nonzero((ia > 0) != (ib > 0))[0][:10])     # a dropped late row: 0 against > 0
        ok = (sa == 0) & (sb == 0)
        assert ok.mean() > 0.99
        assert ua[ok].min() >= -1e-7 and ua[ok].max() <= 22.0 + 1e-7, (t, ua[ok].min(), ua[ok].max())
        assert np.abs(ua[ok] - ub[ok]).max() < 5e-5 and np.abs(xa[ok] - xb[ok]).max() < 5e-5     # two interior points at tol 1e-11
        b.set_iterate(xa, ua)
        x = sim(x, a.get_u(0), T=0.015, steps=1)
    assert late > 0, "the workload produced no late rows: the test proves nothing"
    a.close(); b.close()
