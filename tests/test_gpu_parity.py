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
    A, Bm, b = s.get_linearisation()
    opts = cref.default_opts()
    for i in range(min(B, 8)):
        Ar, Br, br, _q, _r = cref.linearise(opts, xit[i].copy(), uit[i].copy(), x0[i].copy(), yref[i].copy(), yref_e[i].copy())
        assert np.abs(A[i] - Ar).max() < 1e-12   # FP64, same RK4+VDE arithmetic up to association
        assert np.abs(Bm[i] - Br).max() < 1e-12
        assert np.abs(b[i] - br).max() < 1e-12


@pytest.mark.parametrize("init", ["hover", "acados"])
def test_closed_loop_rti_matches_oracle(oracle, cref, init):
    """20 closed-loop RTI steps of hover regulation for 192 instances (3 waves): iterate,
    controls and QP statistics must match the CPU restatement."""
    from crazyflie_nmpc_amd import BatchSolver, sim
    from crazyflie_nmpc_amd.solver import INIT_ACADOS, INIT_HOVER
    B, N = 192, 50
    x0, yref, yref_e = _problem(oracle, B)
    opts = cref.default_opts()
    if init == "hover":
        xr = np.repeat(x0[:, None, :], N + 1, 1).copy(); ur = np.full((B, N, 4), HOV)
    else:
        xr = np.tile(np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0.0]), (B, N + 1, 1)); ur = np.zeros((B, N, 4))
    s = BatchSolver(B)
    s.set_x0(x0); s.set_yref(yref, yref_e)
    s.init_iterate(INIT_HOVER if init == "hover" else INIT_ACADOS)
    x = x0.copy()
    n_constrained = 0
    for t in range(20):
        s.set_x0(x)
        s.solve(1)
        st, it, rs = s.stats()
        st_r, it_r, rs_r, _ = cref.rti_step(opts, xr, ur, x.copy(), yref, yref_e, nthreads=0)
        xg, ug = s.get_iterate()
        assert (st == 0).all() and (st_r == 0).all(), (t, np.bincount(st), np.bincount(st_r))
        # same algorithm, same tolerances: iteration counts agree except for borderline exits
        assert (np.abs(it - it_r) <= 1).all(), (t, it[it != it_r], it_r[it != it_r])
        same = it == it_r
        assert np.abs(ug[same] - ur[same]).max() < 1e-8, t     # kRPM
        assert np.abs(xg[same] - xr[same]).max() < 1e-8, t
        assert np.abs(ug - ur).max() < 1e-5 and np.abs(xg - xr).max() < 1e-5, t  # borderline exits: tol-level
        n_constrained += int((it > 0).sum())
        u0 = s.get_u(0)
        assert np.abs(u0 - ug[:, 0]).max() == 0.0
        assert np.abs(s.get_x(4) - xg[:, 4]).max() == 0.0
        x = sim(x, u0, T=0.015, steps=1)
        ur[:] = ug; xr[:] = xg  # keep both closed loops on the same trajectory
    assert n_constrained > 50  # the interior-point path was actually exercised
    if init == "hover":
        assert np.abs(x[:, :3] - np.array([0, 0, 0.4])).max() < 0.25  # and the loop regulates


def test_qp_solution_satisfies_kkt_and_matches_dense_oracle(oracle):
    """Independent check: the HIP step equals the dense-QP oracle's step and satisfies the KKT
    conditions of the QP built by the numpy oracle (sympy Jacobians)."""
    from crazyflie_nmpc_amd import BatchSolver
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    B, N = 64, 50
    x0, yref, yref_e = _problem(oracle, B, seed=99, scale=1.5)
    s = BatchSolver(B)
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
        assert np.abs(du - ref["du"]).max() < 5e-6   # IPM tol 1e-8 on complementarity -> ~1e-6 on kRPM
        assert np.abs(dx - ref["dx"]).max() < 5e-6
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
    opts = cref.default_opts()
    xr = np.repeat(x0[:, None, :], 51, 1).copy(); ur = np.full((B, 50, 4), HOV)
    cref.rti_step(opts, xr, ur, x0.copy(), yref, yref_e, nthreads=0)
    xg, ug = s.get_iterate()
    same = it > -1
    assert np.abs(ug - ur).max() < 1e-5
