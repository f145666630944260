"""GPU suite: the FUSED start solve (cfnmpc_opts.start_solve, csrc/cfnmpc_linfactor.hip) against the two-kernel one.

k_linfactor linearises every stage inside the factorisation's wavefront -- same model
(crazyflie_full_model/export_ode_model.py:85-97), same RK4 + forward sensitivities (generate_c_code.py:142), evaluated
with shared sub-expressions and one column per lane -- so gains, feed-forward terms, cost-to-go checkpoints and closed
loops must agree with k_linearise + k_factor to rounding, and with the CPU restatement like the stored path does."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HOV = 15.777730167256925


def _inputs(oracle, B, N, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    x0 = oracle.sample_hover_x0(rng, B, scale=scale)
    yr, ye = oracle.regulation_yref(N, (0.0, 0.0, 0.4))
    return x0, np.repeat(yr[None], B, 0).copy(), np.repeat(ye[None], B, 0).copy()


def _rough_iterate(rng, x0, N):
    """an iterate away from hover: every entry of x_k, u_k exercised (attitude, rates, unequal rotor speeds)"""
    B = x0.shape[0]
    x = np.repeat(x0[:, None, :], N + 1, 1) + 0.15 * rng.standard_normal((B, N + 1, 13))
    x[:, :, 3:7] /= np.linalg.norm(x[:, :, 3:7], axis=2, keepdims=True)
    u = HOV + 2.5 * rng.standard_normal((B, N, 4))
    return x, u


@pytest.mark.parametrize("B,N", [(1, 50), (5, 50), (63, 30), (200, 50), (37, 100), (9, 7)])
def test_fused_factor_matches_two_kernels(oracle, B, N):
    """K, d, checkpoints, status of k_linfactor = those of k_linearise + k_factor (relative 1e-11 on a rough iterate)"""
    from crazyflie_nmpc_amd import BatchSolver, default_opts
    rng = np.random.default_rng(400 + B + N)
    x0, yref, yref_e = _inputs(oracle, B, N, seed=B + N, scale=1.5)
    s = BatchSolver(B, default_opts(N=N))
    xi, ui = _rough_iterate(rng, x0, N)
    s.set_x0(x0); s.set_yref(yref, yref_e); s.set_iterate(xi, ui)
    s.start_factor(1)
    K1, d1, P1, st1 = s.get_factor()
    s.start_factor(2)
    K2, d2, P2, st2 = s.get_factor()
    assert (st1 == 0).all() and (st2 == 0).all()
    assert np.abs(K1).max() > 0.1 and np.abs(d1).max() > 1e-3

    def rel(a, b):
        return np.abs(a - b).max() / max(1.0, np.abs(a).max())
    tol = 1e-11 if N <= 50 else 1e-10   # (rounding grows with the length of the backward recursion)
    assert rel(K1, K2) < tol, rel(K1, K2)
    assert rel(d1, d2) < tol, rel(d1, d2)
    nchk = sum(1 for c in (4, 8, 12, 16, 24, 32) if c < N)
    assert nchk == 0 or np.abs(P1[:, :nchk]).max() > 1.0
    assert rel(P1[:, :nchk], P2[:, :nchk]) < tol
    s.close()


@pytest.mark.parametrize("start_solve", [2, 3])
@pytest.mark.parametrize("active_set", [0, 1])
def test_fused_closed_loop_matches_restatement(oracle, cref, start_solve, active_set):
    """closed loops with the fused factorisation = the CPU restatement, as the stored path (test_gpu_parity.py)"""
    from crazyflie_nmpc_amd import BatchSolver, default_opts, sim
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    B, N = 90, 50
    x0, yref, yref_e = _inputs(oracle, B, N, seed=77, scale=2.0)
    tol = 1e-11
    s = BatchSolver(B, default_opts(tol=tol, active_set=active_set, start_solve=start_solve))
    s.set_x0(x0); s.set_yref(yref, yref_e); s.init_iterate(INIT_HOVER)
    opts = cref.default_opts(N=N, tol=tol, active_set=active_set)
    xr = np.repeat(x0[:, None, :], N + 1, 1).copy(); ur = np.full((B, N, 4), HOV)
    x = x0.copy()
    nqp = 0
    for t in range(6):
        s.set_x0(x); s.solve(1)
        st, it, _ = s.stats()
        st_r, it_r, _, _ = cref.rti_step(opts, xr, ur, x.copy(), yref, yref_e, nthreads=0)
        xg, ug = s.get_iterate()
        assert (st == 0).all() and (st_r == 0).all()
        assert np.abs(ug - ur).max() < 1e-8 and np.abs(xg - xr).max() < 1e-8, t
        if active_set:
            # same solves row by row -- except the rows of a wave that went round AGAIN (a settled solution's tail left the box: the
            # monolithic kernel solves its four rows once more over a longer head, starting from the first attempt's final sets --
            # round 6 -- and a row reports the solves of all its attempts: a few more than the restatement's single run)
            assert (it >= it_r).all() and (it == it_r).mean() > 0.8, (t, it[it != it_r], it_r[it != it_r])
        nqp += int((it > 0).sum())
        x = sim(x, s.get_u(0), T=0.015, steps=1)
        ur[:] = ug; xr[:] = xg
    assert nqp > 0
    s.close()


def test_fused_factor_flags_indefinite_rows(oracle):
    """a NaN iterate must end in status 4 for that vehicle only (as k_factor reports it)"""
    from crazyflie_nmpc_amd import BatchSolver, default_opts
    B, N = 8, 50
    x0, yref, yref_e = _inputs(oracle, B, N, seed=5)
    s = BatchSolver(B, default_opts())
    xi = np.repeat(x0[:, None, :], N + 1, 1).copy(); ui = np.full((B, N, 4), HOV)
    xi[3, 20, 8] = np.nan
    s.set_x0(x0); s.set_yref(yref, yref_e); s.set_iterate(xi, ui)
    s.start_factor(2)
    _, _, _, st = s.get_factor()
    assert st[3] == 4 and (np.delete(st, 3) == 0).all(), st
    s.close()


@pytest.mark.parametrize("B,kick", [(4096, 1.0), (20000, 2.5)])
def test_fused_path_equals_stored_path(oracle, B, kick):
    """whole RTI steps, fused start solve (no stored stage blocks; constrained rows re-linearised into their compact store,
    the interior-point fall-back rows once more after their compaction at B >= 16 384) against the stored-block path with
    the same kernels behind it, in lockstep (same x0 and same iterate at the start of every step): same statuses and solve
    counts; rows solved by active-set solves (exact on both sides) to 1e-9, rows that end in the interior point (two runs
    of an iteration to tol 1e-8 from starts that differ in the last bits) to 1e-5"""
    from crazyflie_nmpc_amd import BatchSolver, default_opts, sim
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    N = 50
    x0, yref, yref_e = _inputs(oracle, B, N, seed=31, scale=kick)
    # (the same scheduling on both sides: monolithic active-set kernel, matrix-free forward sweep)
    sol = [BatchSolver(B, default_opts(start_solve=mode, as_passes=-1, forward_sweep=1)) for mode in (1, 2)]
    for s in sol:
        s.set_x0(x0); s.set_yref(yref, yref_e); s.init_iterate(INIT_HOVER)
    x = x0.copy()
    nipm = nas = 0
    for t in range(4):
        out = []
        for s in sol:
            s.set_x0(x); s.solve(1)
            st, it, _ = s.stats()
            xg, ug = s.get_iterate()
            out.append((st, it, xg, ug))
        (st1, it1, x1, u1), (st2, it2, x2, u2) = out
        assert np.array_equal(st1, st2), (t, np.bincount(st1), np.bincount(st2))
        ok = st1 == 0
        assert ok.mean() > 0.99
        assert (it1[ok] == it2[ok]).mean() > 0.999
        err = np.maximum(np.abs(u1 - u2).max(axis=(1, 2)), np.abs(x1 - x2).max(axis=(1, 2)))
        assert err[ok].max() < 1e-5, (t, err[ok].max())
        assert (err[ok] < 1e-9).mean() > 0.97, (t, (err[ok] < 1e-9).mean())
        unc = ok & (it1 == 0)
        # the unconstrained rows never meet the QP kernels: gains agree to 1e-11 relative, the roll-out of a vehicle several
        # units away from its iterate (kick 2.5) amplifies that to a few 1e-9
        assert err[unc].max() < (1e-10 if kick <= 1 else 2e-8)
        nas += int((ok & (it1 > 0) & (it1 <= 12)).sum()); nipm += int((it1 > 12).sum())
        sol[1].set_iterate(x1, u1)             # lockstep: both start the next step from the stored path's iterate
        x = sim(x, u1[:, 0, :].copy(), T=0.015, steps=1)
        if t == 1:   # a disturbance in mid-flight: constrained rows with long heads, fall-back rows
            x = x + kick * np.random.default_rng(9).standard_normal(x.shape) * np.array([.2, .2, .2, .05, .05, .05, .05, .5, .5, .5, 1, 1, 1])
            x[:, 3:7] /= np.linalg.norm(x[:, 3:7], axis=1, keepdims=True)
    assert nas > 0
    if kick > 2:
        assert nipm > 0     # the compacted fall-back list (k_ipm_list + second k_linearise_clist) was exercised
    for s in sol:
        s.close()


def test_start_solve_option_validation_and_per_stage_boxes(oracle):
    """start_solve = 2 is refused together with what reads the stored blocks (cond_N2, forward_sweep = 2, as_passes = -3); a fused solver that later gets per-stage boxes (cfnmpc_set_box_stages) switches to the stored-block
    kernels at launch time and equals a stored-block solver with the same boxes."""
    from crazyflie_nmpc_amd import BatchSolver, default_opts
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    for bad in (dict(cond_N2=10), dict(forward_sweep=2), dict(as_passes=-3), dict(start_solve=4)):
        kw = dict(start_solve=2); kw.update(bad)
        with pytest.raises(Exception):
            BatchSolver(16, default_opts(**kw))
    B, N = 300, 50
    x0, yref, yref_e = _inputs(oracle, B, N, seed=3, scale=2.0)
    lb = np.zeros((B, N, 4)); ub = np.full((B, N, 4), 22.0)
    ub[:, 3:9, :] = 18.0                       # a tighter box on a few stages
    lb[:, 0, :] = ub[:, 0, :] = HOV            # and the FIXED_U0 pattern: stage 0 pinned
    outs = []
    for mode in (1, 2):
        s = BatchSolver(B, default_opts(start_solve=mode, as_passes=-1, forward_sweep=1))
        s.set_x0(x0); s.set_yref(yref, yref_e); s.init_iterate(INIT_HOVER)
        s.solve(1)                              # (mode 2: one fused step first)
        s.set_box_stages(lb, ub)
        s.set_x0(x0); s.init_iterate(INIT_HOVER); s.solve(2)
        st, it, _ = s.stats()
        outs.append((st, it) + s.get_iterate())
        s.set_box_stages(None, None)            # back to the scalar box: the fused kernels again
        s.solve(1)
        assert (s.stats()[0] == 0).mean() > 0.99
        s.close()
    (st1, it1, x1, u1), (st2, it2, x2, u2) = outs
    assert np.array_equal(st1, st2) and (st1 == 0).mean() > 0.99
    ok = st1 == 0
    assert np.abs(u1[ok] - u2[ok]).max() < 1e-8 and np.abs(x1[ok] - x2[ok]).max() < 1e-8
    assert np.abs(u1[ok][:, 0, :] - HOV).max() < 1e-9


def test_get_factor_rows_are_the_instances_own(oracle):
    """cfnmpc_debug_get_factor decodes the wave-blocked device layouts (gains [wave][stage][col][inst & 3][4], feed-forward
    terms in the home 4-vector layout, packed checkpoint triangles): every row of a ragged fleet must equal what a solver
    holding that instance alone returns."""
    from crazyflie_nmpc_amd import BatchSolver, default_opts
    B, N = 7, 50
    rng = np.random.default_rng(11)
    x0, yref, yref_e = _inputs(oracle, B, N, seed=12, scale=1.5)
    xi, ui = _rough_iterate(rng, x0, N)
    s = BatchSolver(B, default_opts(N=N))
    s.set_x0(x0); s.set_yref(yref, yref_e); s.set_iterate(xi, ui)
    s.start_factor(1)
    K, d, P, st = s.get_factor()
    s.close()
    assert (st == 0).all()
    for i in (0, 3, 4, 6):
        s1 = BatchSolver(1, default_opts(N=N))
        s1.set_x0(x0[i:i + 1]); s1.set_yref(yref[i:i + 1], yref_e[i:i + 1]); s1.set_iterate(xi[i:i + 1], ui[i:i + 1])
        s1.start_factor(1)
        K1, d1, P1, st1 = s1.get_factor()
        s1.close()
        assert np.array_equal(K[i], K1[0]) and np.array_equal(d[i], d1[0]) and np.array_equal(P[i], P1[0]), i
        assert np.abs(d1).max() > 1e-3
    # the checkpoints come back symmetric (one triangle is stored)
    assert np.array_equal(P, np.swapaxes(P, 2, 3))
