"""GPU parity: HIP path (through the C-ABI) vs the CPU restatement on identical seeded inputs.
FP64 tolerances are written at each assert.  PARITY UNPINNED w.r.t. acados itself (no oracle
from the reference exists); see oracle/ headers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HOV = 15.777730167256925


def _problem(oracle, B, N=50, seed=20200102, scale=1.0, target=(0.0, 0.0, 0.4)):
    rng = np.random.default_rng(seed)
    x0 = oracle.sample_hover_x0(rng, B, scale=scale)
    yr, ye = oracle.regulation_yref(N, target)
    yref = np.repeat(yr[None], B, 0).copy()
    yref_e = np.repeat(ye[None], B, 0).copy()
    return x0, yref, yref_e


def test_sim_matches_oracle(oracle, cref):
    from crazyflie_nmpc_amd import sim
    rng = np.random.default_rng(1)
    x = oracle.sample_hover_x0(rng, 257)
    u = rng.uniform(0, 22, (257, 4))
    got = sim(x, u, T=0.06, steps=4)
    want = cref.sim(x, u, 0.06, 4)
    assert np.abs(got - want).max() < 1e-12
    one = sim(x, u, T=0.015, steps=1)
    assert np.abs(one - np.stack([oracle.rk4(x[i], u[i]) for i in range(257)])).max() < 1e-12


@pytest.mark.parametrize("B", [1, 63, 200])
def test_linearisation_matches_oracle(oracle, cref, B):
    from crazyflie_nmpc_amd import BatchSolver
    x0, yref, yref_e = _problem(oracle, B)
    rng = np.random.default_rng(5)
    N = 50
    xit = np.repeat(x0[:, None, :], N + 1, 1) + 0.05 * rng.standard_normal((B, N + 1, 13))
    uit = rng.uniform(2, 20, (B, N, 4))
    s = BatchSolver(B)
    s.set_x0(x0); s.set_yref(yref, yref_e); s.set_iterate(xit, uit)
    s.linearise_only()
    opts = cref.default_opts()
    for _rep in (0,):
        A, Bm, b = s.get_linearisation()
        for i in list(range(min(B, 6))) + [B - 1]:
            Ar, Br, br, _q, _r = cref.linearise(opts, xit[i].copy(), uit[i].copy(), x0[i].copy(), yref[i].copy(), yref_e[i].copy())
            assert np.abs(A[i] - Ar).max() < 1e-12   # FP64, same RK4+VDE arithmetic up to association
            assert np.abs(Bm[i] - Br).max() < 1e-12
            assert np.abs(b[i] - br).max() < 1e-12


@pytest.mark.parametrize("init,active_horizon,tol,active_set",
                         [("hover", 0, 1e-8, 0), ("acados", 0, 1e-8, 0), ("hover", 1, 1e-11, 0), ("hover", 1, 1e-8, 0),
                          ("hover", 0, 1e-8, 1), ("hover", 1, 1e-8, 1), ("acados", 0, 1e-8, 1)])
def test_closed_loop_rti_matches_oracle(oracle, cref, init, active_horizon, tol, active_set):
    """20 closed-loop RTI steps of hover regulation for 192 instances (48 waves): iterate,
    controls and QP statistics against the CPU restatement.

    active_horizon=0 runs exactly the oracle's algorithm (all N stages in every interior-point
    sweep): FP64 agreement 1e-8.  active_horizon=1 restricts the interior-point sweeps to the
    head of the horizon (exact reformulation, different central path): both solvers converge
    to the same unique QP solution, so agreement is set by the QP tolerance (tested at 1e-11
    -> 1e-8 on the iterate, and at the default 1e-8 -> 1e-5).  active_set=1 (the engine's default)
    solves the QP by primal-dual active-set iterations on both sides (the restatement's as_solve):
    exact solutions, so the agreement is FP64-level whatever `tol` and head, and with full-horizon
    sweeps the number of solves coincides instance by instance."""
    from crazyflie_nmpc_amd import BatchSolver, sim, default_opts
    from crazyflie_nmpc_amd.solver import INIT_ACADOS, INIT_HOVER
    B, N = 192, 50
    x0, yref, yref_e = _problem(oracle, B)
    opts = cref.default_opts(tol=tol, active_set=active_set)
    if init == "hover":
        xr = np.repeat(x0[:, None, :], N + 1, 1).copy(); ur = np.full((B, N, 4), HOV)
    else:
        xr = np.tile(np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0.0]), (B, N + 1, 1)); ur = np.zeros((B, N, 4))
    s = BatchSolver(B, default_opts(active_horizon=active_horizon, tol=tol, active_set=active_set))
    s.set_x0(x0); s.set_yref(yref, yref_e)
    s.init_iterate(INIT_HOVER if init == "hover" else INIT_ACADOS)
    x = x0.copy()
    n_constrained = 0
    short_heads = 0
    strict = 1e-8 if (active_horizon == 0 or tol <= 1e-11) else 1e-5
    if active_set and init == "hover":
        strict = 1e-8      # both sides exact (no instance falls back to the interior point on this workload)
    elif active_set:
        strict = 5e-4      # cold start: some instances fall back to the interior point at `tol`
    for t in range(20):
        s.set_x0(x)
        s.solve(1)
        st, it, rs = s.stats()
        st_r, it_r, rs_r, _ = cref.rti_step(opts, xr, ur, x.copy(), yref, yref_e, nthreads=0)
        xg, ug = s.get_iterate()
        assert (st == 0).all() and (st_r == 0).all(), (t, np.bincount(st), np.bincount(st_r))
        assert ((it > 0) == (it_r > 0)).all()      # same instances needed the interior-point method
        if active_horizon == 0 and active_set and init == "hover":
            assert np.array_equal(it, it_r), (t, it[it != it_r], it_r[it != it_r])   # same solves, pass by pass
            assert np.abs(ug - ur).max() < strict and np.abs(xg - xr).max() < strict, (t, np.abs(ug - ur).max())
        elif active_horizon == 0 and not active_set:
            # same algorithm, same tolerances: iteration counts agree except for borderline exits
            assert (np.abs(it - it_r) <= 1).all(), (t, it[it != it_r], it_r[it != it_r])
            same = it == it_r
            assert np.abs(ug[same] - ur[same]).max() < 1e-8, t     # kRPM
            assert np.abs(xg[same] - xr[same]).max() < 1e-8, t
            assert np.abs(ug - ur).max() < 1e-5 and np.abs(xg - xr).max() < 1e-5, t  # borderline exits: tol-level
        elif active_set and init != "hover":
            # cold start: the same algorithm on both sides (active-set solves, interior point at `tol` for
            # the instances whose active set does not settle) -- wherever the solve counts coincide the
            # iterates agree at FP64 level; an interior-point fall-back that exits one iteration apart
            # differs at the level of its tolerance
            same = it == it_r
            assert same.mean() > 0.9, (t, same.mean())
            assert np.abs(ug[same] - ur[same]).max() < 1e-8 and np.abs(xg[same] - xr[same]).max() < 1e-8, t
            assert np.abs(ug - ur).max() < strict and np.abs(xg - xr).max() < strict, (t, np.abs(ug - ur).max())
        else:
            assert np.abs(ug - ur).max() < strict and np.abs(xg - xr).max() < strict, (t, np.abs(ug - ur).max())
            if active_set and init == "hover":
                assert it.max() <= 6    # active-set solves, no fall-back to the interior point on this workload
            short_heads += int((s.heads()[it > 0] < N).sum())
        n_constrained += int((it > 0).sum())
        u0 = s.get_u(0)
        assert np.abs(u0 - ug[:, 0]).max() == 0.0
        assert np.abs(s.get_x(4) - xg[:, 4]).max() == 0.0
        x = sim(x, u0, T=0.015, steps=1)
        ur[:] = ug; xr[:] = xg  # keep both closed loops on the same trajectory
    assert n_constrained > 50  # the interior-point path was actually exercised
    if active_horizon:
        assert short_heads > 50  # ... and mostly on a shortened horizon
    if init == "hover":
        assert np.abs(x[:, :3] - np.array([0, 0, 0.4])).max() < 0.25  # and the loop regulates


@pytest.mark.parametrize("active_set,tol,bound", [(0, 1e-12, 5e-6), (0, 1e-8, 5e-4), (1, 1e-8, 5e-9)])
def test_qp_solution_matches_dense_oracle(oracle, active_set, tol, bound):
    """Independent check: the HIP step equals the step of the dense-QP oracle on the QP built by
    the numpy oracle (sympy Jacobians).  An interior-point solution sits on the central path:
    for a (nearly) degenerate bound slack ~ multiplier ~ sqrt(mu), so the primal error is bounded
    by ~sqrt(tol) -- 1e-6 at tol 1e-12 and 1e-4 at the default 1e-8 (same property as HPIPM).  The
    active-set solves (the engine's default) are exact: they meet the dense oracle at its own accuracy."""
    from crazyflie_nmpc_amd import BatchSolver, default_opts
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    B, N = 64, 50
    x0, yref, yref_e = _problem(oracle, B, seed=99, scale=1.5)
    s = BatchSolver(B, default_opts(tol=tol, active_set=active_set))
    s.set_x0(x0); s.set_yref(yref, yref_e); s.init_iterate(INIT_HOVER)
    s.solve(1)
    xg, ug = s.get_iterate()
    st, it, _ = s.stats()
    assert (st == 0).all()
    checked = 0
    for i in np.argsort(-it)[:6]:
        xbar = np.repeat(x0[i][None], N + 1, 0); ubar = np.full((N, 4), HOV)
        qp = oracle.build_qp(xbar, ubar, x0[i], yref[i], yref_e[i])
        ref = oracle.solve_qp_dense(qp)
        du = ug[i] - ubar
        dx = xg[i] - xbar
        assert np.abs(du - ref["du"]).max() < bound
        assert np.abs(dx - ref["dx"]).max() < bound
        checked += it[i] > 0
    assert checked >= 1


def test_ragged_batch_and_status(oracle, cref):
    """Batch not a multiple of the wave size; hard instances (3x perturbation) still converge."""
    from crazyflie_nmpc_amd import BatchSolver
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    B = 97
    x0, yref, yref_e = _problem(oracle, B, seed=4, scale=3.0)
    s = BatchSolver(B)
    s.set_x0(x0); s.set_yref(yref, yref_e); s.init_iterate(INIT_HOVER)
    s.solve(1)
    st, it, rs = s.stats()
    assert (st == 0).all() and it.max() <= 30 and np.nanmax(rs) <= 1e-8
    opts = cref.default_opts(active_set=1)
    xr = np.repeat(x0[:, None, :], 51, 1).copy(); ur = np.full((B, 50, 4), HOV)
    cref.rti_step(opts, xr, ur, x0.copy(), yref, yref_e, nthreads=0)
    xg, ug = s.get_iterate()
    # exact active-set solutions on both sides (the engine sweeps active horizons, the restatement
    # the full one: same unique solution)
    assert np.abs(ug - ur).max() < 1e-8 and np.abs(xg - xr).max() < 1e-8


def test_active_set_solves_match_oracle_exactly(oracle):
    """The engine's default QP method (primal-dual active-set solves on the Riccati
    factorisation) against its CPU twin oracle.pdas_dense on QPs built by the numpy oracle (sympy
    Jacobians): the same number of solves for every instance -- the classifications coincide pass
    by pass -- and the same solution to 1e-9 (both are exact; nothing of an interior point's
    central-path error is left).  Full-horizon sweeps so that both solve the same problem."""
    from crazyflie_nmpc_amd import BatchSolver, default_opts
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    B, N = 48, 50
    x0, yref, yref_e = _problem(oracle, B, seed=321, scale=2.0)
    s = BatchSolver(B, default_opts(active_horizon=0, active_set=1))
    s.set_x0(x0); s.set_yref(yref, yref_e); s.init_iterate(INIT_HOVER)
    s.solve(1)
    st, it, _ = s.stats()
    xg, ug = s.get_iterate()
    assert (st == 0).all()
    n_act = 0
    for i in range(B):
        xbar = np.repeat(x0[i][None], N + 1, 0); ubar = np.full((N, 4), HOV)
        qp = oracle.build_qp(xbar, ubar, x0[i], yref[i], yref_e[i])
        sol = oracle.pdas_dense(qp)
        assert sol["converged"]
        assert it[i] == sol["solves"], (i, it[i], sol["solves"])
        assert np.abs(ug[i] - (ubar + sol["du"])).max() < 1e-9 and np.abs(xg[i] - (xbar + sol["dx"])).max() < 1e-9, i
        n_act += sol["solves"] > 0
    assert n_act >= 10


@pytest.mark.parametrize("scale,init", [(3.0, "hover"), (1.0, "acados")])
def test_active_set_under_heavy_saturation(oracle, cref, scale, init):
    """Stress of the default QP method: large perturbations (many inputs at both bounds over long
    heads) and the cold acados start (iterate far from the measurement: the active-set iteration
    does not always settle and hands over to the interior point).  Whatever route an instance
    takes, it must end at the restatement's solution, inside the box, with status 0.
    Rows that fall back to the interior point run it over the FULL horizon (round 6: the restatement's own algorithm, one
    attempt instead of a run over the classified head + another after its tail check failed): head = N, and the iteration
    counts are the restatement's up to borderline exits."""
    from crazyflie_nmpc_amd import BatchSolver, sim, default_opts
    from crazyflie_nmpc_amd.solver import INIT_ACADOS, INIT_HOVER
    B, N, tol = 256, 50, 1e-10
    x0, yref, yref_e = _problem(oracle, B, seed=4242, scale=scale)
    opts = cref.default_opts(tol=tol)
    if init == "hover":
        xr = np.repeat(x0[:, None, :], N + 1, 1).copy(); ur = np.full((B, N, 4), HOV)
    else:
        xr = np.tile(np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0.0]), (B, N + 1, 1)); ur = np.zeros((B, N, 4))
    s = BatchSolver(B, default_opts(tol=tol))
    s.set_x0(x0); s.set_yref(yref, yref_e)
    s.init_iterate(INIT_HOVER if init == "hover" else INIT_ACADOS)
    x = x0.copy()
    n_con = 0
    for t in range(4):
        s.set_x0(x); s.solve(1)
        st, it, rs = s.stats()
        st_r, it_r, rs_r, _ = cref.rti_step(opts, xr, ur, x.copy(), yref, yref_e, nthreads=0)
        xg, ug = s.get_iterate()
        assert (st == 0).all() and (st_r == 0).all(), (t, np.bincount(st), np.bincount(st_r))
        assert ((it > 0) == (it_r > 0)).all()
        assert ug.min() > -1e-9 and ug.max() < 22.0 + 1e-9
        assert np.abs(ug - ur).max() < 2e-5 and np.abs(xg - xr).max() < 2e-5, (t, np.abs(ug - ur).max())
        fb = (it > 0) & (rs > 0)                       # interior-point rows (active-set rows report res = 0 exactly)
        assert (s.heads()[fb] == N).all(), (t, s.heads()[fb])
        both = fb & (rs_r > 0)
        assert (np.abs(it[both] - it_r[both]) <= 1).all(), (t, it[both], it_r[both])
        n_con += int((it > 0).sum())
        x = sim(x, s.get_u(0), T=0.015, steps=1)
        ur[:] = ug; xr[:] = xg
    assert n_con > B // 2


def test_fallback_rows_run_the_full_horizon(oracle, cref):
    """Heavily disturbed closed loop (3 x the bench's kicks, a cohort re-kicked every step): about one row in a hundred falls back
    to the interior point -- its active set does not settle within the cap, or it skips the iteration (as_skip_viol).  Those rows
    run the interior point over the FULL horizon in one attempt (head = N; up to round 5: over the classified head first and
    over the full horizon after its tail check had failed -- 99 % of them), which is the restatement's own algorithm: the SAME
    iteration counts, iterates at FP64 level times the conditioning (measured: <= 1e-6 at equal residuals)."""
    from crazyflie_nmpc_amd import BatchSolver, sim, default_opts
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    B, N, KP = 768, 50, 8
    rng = np.random.default_rng(606)
    x = oracle.sample_hover_x0(rng, B, scale=3.0)
    yr, yre = oracle.regulation_yref(N, (0, 0, 0.4))
    yref = np.repeat(yr[None], B, 0).copy(); yref_e = np.repeat(yre[None], B, 0).copy()
    xr = np.repeat(x[:, None, :], N + 1, 1).copy(); ur = np.full((B, N, 4), HOV)
    opts = cref.default_opts(active_set=1)
    s = BatchSolver(B, default_opts())
    s.set_x0(x); s.set_yref(yref, yref_e); s.init_iterate(INIT_HOVER)
    n_fb = n_same = n_mixed = n_con = 0
    for t in range(6):
        c0 = (t % KP) * (B // KP)
        x[c0:c0 + B // KP] = oracle.sample_hover_x0(rng, B // KP, scale=3.0)
        s.set_x0(x); s.solve(1)
        st, it, rs = s.stats()
        st_r, it_r, rs_r, _ = cref.rti_step(opts, xr, ur, x.copy(), yref, yref_e, nthreads=0)
        xg, ug = s.get_iterate()
        ok = (st == 0) & (st_r == 0)
        assert ok.mean() > 0.98, (t, np.bincount(st), np.bincount(st_r))
        assert ((it > 0) == (it_r > 0))[ok].all()
        fb = ok & (it > 0) & (rs > 0)                  # interior-point rows (active-set rows report res = 0 exactly)
        assert (s.heads()[fb] == N).all(), (t, s.heads()[fb])
        both = fb & (rs_r > 0)
        same = both & (it == it_r)
        err = np.maximum(np.abs(ug - ur).reshape(B, -1).max(1), np.abs(xg - xr).reshape(B, -1).max(1))
        assert err[same].max(initial=0.0) < 5e-6, (t, err[same].max())
        assert err[ok].max() < 5e-4, (t, err[ok].max())        # (a row that took the other route, a borderline exit: tolerance level)
        n_fb += int(both.sum()); n_same += int(same.sum()); n_con += int((ok & (it > 0)).sum())
        n_mixed += int((ok & (it > 0) & ((rs > 0) != (rs_r > 0))).sum())
        x = sim(x, s.get_u(0), T=0.015, steps=1)
        ur[:] = ug; xr[:] = xg
    assert n_fb >= 10, n_fb              # the fall-back was exercised ...
    assert n_same >= 0.9 * n_fb, (n_same, n_fb)   # ... with the restatement's iteration counts
    assert n_mixed <= 0.01 * n_con, (n_mixed, n_con)


@pytest.mark.parametrize("active_horizon", [0, 1])
def test_warm_started_active_set_matches_restatement(oracle, cref, active_horizon):
    """cfnmpc_opts.as_warm: 20 closed-loop steps with 2x kicks (most vehicles constrained for several consecutive steps), the
    engine and the C restatement both starting each first solve from the union of the previous step's final set and today's
    violations.  Full-horizon sweeps: the solve counts agree vehicle by vehicle and step by step (the two implementations
    walk through the same classifications); active-horizon sweeps: same solutions.  And the warm-started engine reaches the
    iterates of the cold-started one (exactness does not depend on the start)."""
    from crazyflie_nmpc_amd import BatchSolver, sim, default_opts
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    B, N = 192, 50
    rng = np.random.default_rng(99)
    x = oracle.sample_hover_x0(rng, B, scale=2.0)
    yr, ye = oracle.regulation_yref(N, (0.0, 0.0, 0.4))
    yref = np.repeat(yr[None], B, 0).copy(); yref_e = np.repeat(ye[None], B, 0).copy()
    opts = cref.default_opts(active_set=1, as_warm=1)
    warm = cref.warm_state(B, N)
    xr = np.repeat(x[:, None, :], N + 1, 1).copy(); ur = np.full((B, N, 4), HOV)
    s = BatchSolver(B, default_opts(active_horizon=active_horizon, as_warm=1))
    c = BatchSolver(B, default_opts(active_horizon=active_horizon))
    for q in (s, c):
        q.set_x0(x); q.set_yref(yref, yref_e); q.init_iterate(INIT_HOVER)
    kicks = oracle.sample_hover_x0(rng, 9 * 20, scale=2.0).reshape(20, 9, 13)
    warm_rows = n_constrained = 0
    for t in range(20):
        x[t * 9:t * 9 + 9] = kicks[t]
        s.set_x0(x); s.solve(1)
        xc, uc = s.get_iterate()          # the cold engine and the restatement continue from the warm engine's iterate
        c.set_x0(x); c.solve(1)
        st, it, rs = s.stats(); st_c, it_c, rs_c = c.stats()
        warm_rows += int((warm[1] != 0).sum())
        st_r, it_r, rs_r, _ = cref.rti_step(opts, xr, ur, x.copy(), yref, yref_e, nthreads=0, warm=warm)
        xg, ug = s.get_iterate()
        xk, uk = c.get_iterate()
        ok = (st == 0) & (st_r == 0) & (st_c == 0)
        assert ok.mean() > 0.97 and np.array_equal(st, st_r)
        assert np.array_equal(it > 0, it_r > 0) and np.array_equal(it > 0, it_c > 0)
        as_rows = ok & (rs == 0.0) & (rs_r == 0.0) & (rs_c == 0.0)    # settled by active-set solves on every side (an interior-point row reports its residual, > 0,
        #                                                                whatever its iteration count: twelve iterations are not twelve solves)
        if active_horizon == 0:
            assert np.array_equal(it[as_rows], it_r[as_rows]), (t, it[as_rows], it_r[as_rows])
        assert np.abs(ug[as_rows] - ur[as_rows]).max() < 1e-7 and np.abs(xg[as_rows] - xr[as_rows]).max() < 1e-7, t
        # warm and cold engines: the same solution, reached through different sequences of classifications (at 2x kicks some
        # QPs are conditioned badly enough that two exact solves differ by 1e-6: tests/test_gpu_delayed_loop.py)
        assert np.abs(ug[as_rows] - uk[as_rows]).max() < 1e-5, t
        assert np.abs(ug[ok] - ur[ok]).max() < 5e-4
        n_constrained += int((it > 0).sum())
        ur[:] = ug; xr[:] = xg
        c.set_iterate(xg, ug)
        x = sim(x, s.get_u(0), T=0.015, steps=1)
    assert n_constrained > 1000 and warm_rows > 500      # the warm start was exercised on most constrained QPs
    s.close(); c.close()
