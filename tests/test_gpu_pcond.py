"""Row a7: partial condensing -- PARTIAL_CONDENSING_HPIPM of the reference's
solver plan (generate_c_code.py:140) -- as cfnmpc_opts.cond_N2.  HPIPM is not under /root/reference
(PARITY UNPINNED, as for the whole optimiser); the checks are
  * the condensed blocks against the numpy restatement of the published algorithm
    (oracle.partial_condense) on a generic iterate,
  * the primal solution for N2 in {25, 10, 5} (+ an uneven split) against the exact solutions of the
    dense oracle committed in tests/golden/qp.npz and against the uncondensed path (N2 = N),
  * closed loops against the CPU restatement."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HOV = 15.777730167256925
N = 50


@pytest.mark.parametrize("N2", [25, 10, 5, 7])
def test_condensed_blocks_match_oracle(oracle, N2):
    """k_pcond == oracle.partial_condense (H, D of every block) on a perturbed iterate with a
    tracking reference: 1e-10 relative to the block's largest entry."""
    from crazyflie_nmpc_amd import BatchSolver, default_opts
    rng = np.random.default_rng(11)
    B = 6
    x0 = oracle.sample_hover_x0(rng, B, scale=1.0)
    xit = np.repeat(x0[:, None, :], N + 1, 1) + 0.02 * rng.standard_normal((B, N + 1, 13))
    uit = HOV + 0.3 * rng.standard_normal((B, N, 4))
    yref = np.zeros((B, N, 17)); yref[:, :, :3] = rng.uniform(-0.5, 0.5, (B, N, 3)); yref[:, :, 3] = 1; yref[:, :, 13:] = HOV
    yref_e = yref[:, -1, :13].copy()
    s = BatchSolver(B, default_opts(cond_N2=N2))
    s.set_x0(x0); s.set_yref(yref, yref_e); s.set_iterate(xit, uit)
    sizes = oracle.block_sizes(N, N2)
    qps = [oracle.build_qp(xit[i], uit[i], x0[i], yref[i], yref_e[i]) for i in range(B)]
    blocks = [oracle.partial_condense(qp, N2) for qp in qps]
    for j in (0, len(sizes) // 2, len(sizes) - 1):
        H, D, m = s.get_condensed(j)
        assert m == sizes[j]
        w = 4 * m + 14
        for i in range(B):
            Ho, Do = blocks[i][j]["H"], blocks[i][j]["D"]
            Hg = H[i].copy(); Hg[w - 1, w - 1] = 0.0
            assert np.abs(Hg - Ho).max() < 1e-10 * max(1.0, np.abs(Ho).max()), (N2, j, i)
            assert np.abs(D[i] - Do).max() < 1e-12, (N2, j, i)


@pytest.mark.parametrize("N2", [25, 10, 5, 7])
def test_condensed_qp_solutions_match_golden_and_uncondensed(N2):
    """Same primal solution for every N2: against the committed exact solutions (unconstrained cases
    1e-9; saturating cases at the interior point's central-path accuracy, tol 1e-12 -> 5e-6, like the
    uncondensed interior point) and against the uncondensed interior point at the same tolerance
    (mathematically the same iterates: 1e-8, equal iteration counts)."""
    from crazyflie_nmpc_amd import BatchSolver, default_opts
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    q = np.load(os.path.join(G, "qp.npz"))
    n = q["x0"].shape[0]
    out = {}
    for key, kw in (("cond", dict(cond_N2=N2)), ("plain", dict(active_set=0, active_horizon=0))):
        s = BatchSolver(n, default_opts(tol=1e-12, **kw))
        s.set_x0(q["x0"]); s.set_yref(np.tile(q["yref"], (n, 1, 1)), np.tile(q["yref_e"], (n, 1))); s.init_iterate(INIT_HOVER)
        s.solve(1)
        st, it, _ = s.stats()
        xg, ug = s.get_iterate()
        assert (st == 0).all(), (key, st)
        out[key] = (xg, ug, it)
    xg, ug, it = out["cond"]
    free = q["n_active"] == 0
    assert ((it > 0) == ~free).all()
    assert np.abs(ug[free] - HOV - q["du"][free]).max() < 1e-9 and np.abs(xg[free] - q["x0"][free, None, :] - q["dx"][free]).max() < 1e-9
    assert np.abs(ug - HOV - q["du"]).max() < 5e-6 and np.abs(xg - q["x0"][:, None, :] - q["dx"]).max() < 5e-6
    xp, up, itp = out["plain"]
    assert np.array_equal(it, itp), (it, itp)
    assert np.abs(ug - up).max() < 1e-8 and np.abs(xg - xp).max() < 1e-8


@pytest.mark.parametrize("N2,B", [(10, 67), (5, 9), (25, 130)])
def test_condensed_closed_loop_matches_restatement(oracle, cref, N2, B):
    """12 closed-loop RTI steps (kicked start, ragged batch) against the CPU restatement's interior
    point: same statuses, inputs to 1e-6 (two interior points at tol 1e-8 on the same QPs)."""
    from crazyflie_nmpc_amd import BatchSolver, default_opts, sim
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    rng = np.random.default_rng(5)
    x = oracle.sample_hover_x0(rng, B, scale=1.5)
    yr, ye = oracle.regulation_yref(N, (0.0, 0.0, 0.4))
    yref = np.repeat(yr[None], B, 0).copy(); yref_e = np.repeat(ye[None], B, 0).copy()
    s = BatchSolver(B, default_opts(cond_N2=N2))
    s.set_x0(x); s.set_yref(yref, yref_e); s.init_iterate(INIT_HOVER)
    xr = np.repeat(x[:, None, :], N + 1, 1).copy(); ur = np.full((B, N, 4), HOV)
    opts = cref.default_opts(active_set=0)
    seen = 0
    for t in range(12):
        s.set_x0(x); s.solve(1)
        st, it, rs = s.stats()
        xg, ug = s.get_iterate()
        st_r, it_r, _, _ = cref.rti_step(opts, xr, ur, x.copy(), yref, yref_e, nthreads=0)
        assert (st == 0).all() and (st_r == 0).all()
        assert ((it > 0) == (it_r > 0)).all()
        seen += int((it > 0).sum())
        assert np.abs(ug - ur).max() < 1e-6 and np.abs(xg - xr).max() < 1e-6, t
        assert np.abs(xg[:, 0, :] - x).max() < 1e-14 and ug.min() >= -1e-7 and ug.max() <= 22 + 1e-7
        xr[:] = xg; ur[:] = ug
        x = sim(x, ug[:, 0, :].copy(), T=0.015, steps=1)
    assert seen > 0


def test_cond_option_validation():
    from crazyflie_nmpc_amd import BatchSolver, default_opts
    from crazyflie_nmpc_amd.solver import CfnmpcError
    for bad in (dict(cond_N2=4), dict(cond_N2=-1), dict(cond_N2=51)):
        with pytest.raises(CfnmpcError):
            BatchSolver(4, default_opts(**bad))      # blocks longer than 10 stages / out of range
    for ok in (0, 50):
        BatchSolver(4, default_opts(cond_N2=ok)).close()    # both mean "no condensing"


def test_condensed_mixed_horizon_fleet(oracle):
    """cond_N2 reaches every bucket of a mixed-horizon fleet (blocks of 3 / 5 / 10 stages for N = 30 / 50 /
    100 with cond_N2 = 10) and gives the uncondensed fleet's controls; a cond_N2 that would need blocks
    longer than 10 stages in some bucket is refused at creation."""
    from crazyflie_nmpc_amd.fleet import MixedHorizonFleet
    from crazyflie_nmpc_amd.solver import CfnmpcError, INIT_HOVER
    rng = np.random.default_rng(61)
    B = 41
    hz = rng.choice([30, 50, 100], size=B)
    x0 = oracle.sample_hover_x0(rng, B, scale=1.5)
    res = []
    for kw in (dict(cond_N2=10), dict(active_set=0, active_horizon=0)):
        f = MixedHorizonFleet(hz, **kw)
        f.set_regulation(np.tile([0.0, 0.0, 0.4], (B, 1)), HOV)
        f.set_x0(x0); f.init_iterate(INIT_HOVER); f.solve(1)
        st, it, _ = f.stats()
        assert (st == 0).all() and (it > 0).any()
        res.append((f.get_u(0), f.get_u(1), f.get_x(4), it))
        f.close()
    (u0c, u1c, x4c, itc), (u0p, u1p, x4p, itp) = res
    assert np.array_equal(itc, itp)
    assert np.abs(u0c - u0p).max() < 1e-6 and np.abs(u1c - u1p).max() < 1e-6 and np.abs(x4c - x4p).max() < 1e-6
    with pytest.raises(CfnmpcError):
        MixedHorizonFleet(hz, cond_N2=5)          # N = 100 would need 20-stage blocks
    # a bucket with no more stages than cond_N2 simply runs uncondensed (cond_N2 >= N means "none")
    hz2 = rng.choice([8, 30, 50], size=B)
    f = MixedHorizonFleet(hz2, cond_N2=10)
    f.set_regulation(np.tile([0.0, 0.0, 0.4], (B, 1)), HOV)
    f.set_x0(x0); f.init_iterate(INIT_HOVER); f.solve(1)
    assert (f.stats()[0] == 0).all()
    f.close()


@pytest.mark.parametrize("seed", list(range(8)))
def test_condensed_randomised_horizons_blocks_and_options(oracle, cref, seed):
    """Seeded fuzz: horizon, number of blocks (uneven splits included, block lengths 1..10), interval,
    weights, box and batch at once -- the condensed path against the CPU restatement's interior point
    (both at tol 1e-11), two closed-loop steps."""
    from crazyflie_nmpc_amd import BatchSolver, default_opts
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    rng = np.random.default_rng(8800 + seed)
    Nh = int(rng.integers(6, 65))
    lo = -(-Nh // 10)                                   # blocks of at most 10 stages
    N2 = int(rng.integers(lo, Nh))                      # lo <= N2 < N
    dt = float(rng.uniform(0.008, 0.025))
    B = int(rng.integers(1, 90))
    d = default_opts()
    W = np.array(list(d.W)) * np.exp(rng.uniform(np.log(0.5), np.log(2.0), 17))
    WN = np.array(list(d.WN)) * np.exp(rng.uniform(np.log(0.5), np.log(2.0), 13))
    u_min, u_max = float(rng.uniform(0.0, 6.0)), float(rng.uniform(19.0, 24.0))
    x0 = oracle.sample_hover_x0(rng, B, scale=float(rng.uniform(0.5, 2.0)))
    yr, ye = oracle.regulation_yref(Nh, (0.1, -0.1, 0.5))
    yref = np.repeat(yr[None], B, 0).copy(); yref_e = np.repeat(ye[None], B, 0).copy()
    kw = dict(N=Nh, dt=dt, W=W, WN=WN, u_min=u_min, u_max=u_max, tol=1e-11)
    s = BatchSolver(B, default_opts(cond_N2=N2, **kw))
    s.set_x0(x0); s.set_yref(yref, yref_e); s.init_iterate(INIT_HOVER)
    xr = np.repeat(x0[:, None, :], Nh + 1, 1).copy(); ur = np.full((B, Nh, 4), HOV)
    opts = cref.default_opts(active_set=0, **kw)
    x = x0.copy()
    for t in range(2):
        s.set_x0(x); s.solve(1)
        st, it, _ = s.stats()
        xg, ug = s.get_iterate()
        st_r, it_r, _, _ = cref.rti_step(opts, xr, ur, x.copy(), yref, yref_e, nthreads=0)
        assert (st == st_r).all(), (seed, Nh, N2, B)
        ok = st == 0
        assert ok.mean() > 0.9 and ((it > 0) == (it_r > 0))[ok].all()
        assert np.abs(ug - ur)[ok].max() < 5e-6 and np.abs(xg - xr)[ok].max() < 5e-6, (seed, Nh, N2, B)
        xr[:] = xg; ur[:] = ug
        x = xg[:, 1, :].copy()


def test_start_solve_accessors_refuse_condensed_solvers():
    """cfnmpc_debug_start_factor / _get_factor run kernels that index the home 4-vectors in the wave-blocked layout; a
    partial-condensing solver keeps them instance-major and must be refused (CFNMPC_EINVAL), not silently mis-addressed.
    On a short horizon the accessor leaves the checkpoints no kernel fills (stage >= N) untouched."""
    from crazyflie_nmpc_amd import BatchSolver, default_opts
    from crazyflie_nmpc_amd.solver import CfnmpcError, INIT_HOVER
    from crazyflie_nmpc_amd.synthetic import regulation_row, sample_hover_x0
    s = BatchSolver(9, default_opts(cond_N2=10))
    with pytest.raises(CfnmpcError):
        s.start_factor(1)
    with pytest.raises(CfnmpcError):
        s.get_factor()
    s.close()
    B, N = 9, 20
    s = BatchSolver(B, default_opts(N=N))
    row = regulation_row()
    s.set_yref(np.tile(row, (B, N, 1)), np.tile(row[:13], (B, 1)))
    s.set_x0(sample_hover_x0(np.random.default_rng(3), B)); s.init_iterate(INIT_HOVER)
    s.start_factor(1)
    K, d, Pc, st = s.get_factor()          # Pc starts as zeros in the wrapper
    assert (st == 0).all() and np.isfinite(K).all()
    assert np.abs(Pc[:, :4]).max() > 0 and not Pc[:, 4:].any()     # checkpoints at stages 4, 8, 12, 16 | 24, 32 >= N
    s.close()
