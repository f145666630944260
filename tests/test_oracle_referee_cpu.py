"""The extended-precision referee (oracle.solve_qp_refined) and what it established in round 6 about the one fuzz row that
round 5 had given a 2e-6 tolerance (tests/test_gpu_edge_cases.py::test_randomised_options_match_restatement, seed 8): CPU only,
restatement side (the engine's side of the same row is checked on the GPU by that test)."""
import numpy as np

HOV = 15.777730167256925


def _seed8(oracle):
    rng = np.random.default_rng(7700 + 8)
    N = int(rng.integers(2, 71)); dt = float(rng.uniform(0.006, 0.03)); B = int(rng.integers(1, 150))
    W = oracle.W_DIAG * np.exp(rng.uniform(np.log(0.3), np.log(3.0), 17))
    WN = 50.0 * oracle.W_DIAG[:13] * np.exp(rng.uniform(np.log(0.3), np.log(3.0), 13))
    u_min, u_max = float(rng.uniform(0.0, 8.0)), float(rng.uniform(18.0, 24.0))
    rng.integers(0, 2); rng.integers(0, 2); rng.integers(1, 3)
    x0 = oracle.sample_hover_x0(rng, B, scale=float(rng.uniform(0.5, 2.0)))
    yr, ye = oracle.regulation_yref(N, tuple(rng.uniform(-0.3, 0.3, 3) + np.array([0.0, 0.0, 0.5])))
    yref = np.repeat(yr[None], B, 0).copy(); yref_e = np.repeat(ye[None], B, 0).copy()
    yref[:, :, :3] += rng.uniform(-0.05, 0.05, (B, N, 3)); yref[:, :, 13:] += rng.uniform(-0.5, 0.5, (B, N, 4))
    yref_e[:, :3] += rng.uniform(-0.05, 0.05, (B, 3))
    return N, dt, B, W, WN, u_min, u_max, x0, yref, yref_e


def test_referee_agrees_with_the_fp64_solvers_on_a_well_conditioned_qp(oracle):
    rng = np.random.default_rng(3)
    N = 30
    x0 = oracle.sample_hover_x0(rng, 1, scale=2.0)[0]
    yr, ye = oracle.regulation_yref(N, (0.0, 0.0, 0.4))
    qp = oracle.build_qp(np.repeat(x0[None], N + 1, 0), np.full((N, 4), HOV), x0 + 0.01, yr, ye)
    ref = oracle.solve_qp_refined(qp)
    pd = oracle.pdas_dense(qp)
    assert ref["solves"] == pd["solves"] > 0 and np.array_equal(ref["cls"], pd["cls"])
    assert ref["kkt"] < 1e-11 and ref["eps"] < 2e-19
    assert np.abs(pd["du"] - ref["du"]).max() < 1e-9 and np.abs(pd["dx"] - ref["dx"]).max() < 1e-9
    k = oracle.kkt_residual(qp, ref["dx"], ref["du"], *[np.maximum(s * (oracle.condense(qp)[0] @ ref["du"].reshape(-1) + oracle.condense(qp)[1]), 0).reshape(N, 4) * (ref["cls"] == c)
                                                        for s, c in ((1.0, 1), (-1.0, 2))])
    assert k["max"] < 1e-8, k


def test_seed8_row_is_an_interior_point_row_on_both_sides(oracle, cref):
    """The row the fuzz test's tolerance was loosened for: the exact active-set iteration needs 13 solves -- beyond the cap of 12
    that engine and restatement share -- so the restatement finishes it with the interior point (12 ITERATIONS, by coincidence the
    cap's value) and lands 1.9e-6 from the exact solution: central-path accuracy at tol 1e-8, not a conditioning problem -- the
    Riccati form of the solve with the exact set is within 1e-10 of the exact solution of this QP (cond 2e7)."""
    N, dt, B, W, WN, u_min, u_max, x0, yref, yref_e = _seed8(oracle)
    opts = cref.default_opts(N=N, dt=dt, W=W, WN=WN, u_min=u_min, u_max=u_max, active_set=1)
    i = 45
    xp = np.repeat(x0[i:i + 1, None, :], N + 1, 1).copy(); up = np.full((1, N, 4), HOV)
    xr, ur = xp.copy(), up.copy()
    st, it, _, _ = cref.rti_step(opts, xr, ur, x0[i:i + 1].copy(), yref[i:i + 1], yref_e[i:i + 1], nthreads=1)
    A, Bm, b, q, r = cref.linearise(opts, xp[0], up[0], x0[i].copy(), yref[i], yref_e[i])
    qp = oracle.qp_from_blocks(A, Bm, b, q, r, x0[i] - xp[0, 0], W[:13], W[13:], WN, u_min - up[0], u_max - up[0])
    ref = oracle.solve_qp_refined(qp)
    assert ref["kkt"] < 1e-9 and ref["solves"] == 13 and int((ref["cls"] > 0).sum()) == 72
    e_res = max(np.abs(ur[0] - up[0] - ref["du"]).max(), np.abs(xr[0] - xp[0] - ref["dx"]).max())
    assert st[0] == 0 and it[0] == 12 and 5e-8 < e_res < 5e-6, (it, e_res)          # the interior point's accuracy
    # with a cap that admits the 13th solve the restatement IS exact: the dense FP64 twin ...
    pd = oracle.pdas_dense(qp, max_solves=20)
    assert pd["converged"] and pd["solves"] == 13 and np.abs(pd["du"] - ref["du"]).max() < 5e-8
    # ... and the stage-wise Riccati form of the same equality-constrained solve (delta form around the unconstrained minimiser,
    # fixed inputs through 1e30 on their diagonal and b_eff = B c: oracle/cfnmpc_ref.c as_solve, the engine's sweep_factor_as)
    cls = ref["cls"]
    Rd = np.tile(qp.Rd, (N, 1))
    K, Sinv, d = oracle._riccati_factor(qp, Rd, qp.r, absolute=True)
    _, v0 = oracle._riccati_forward(qp, K, d, absolute=True)
    c = np.where(cls == 1, qp.lb - v0, np.where(cls == 2, qp.ub - v0, 0.0))
    import copy
    qd = copy.deepcopy(qp)
    qd.b = np.einsum("kia,ka->ki", qp.B, c); qd.q = np.zeros_like(qp.q); qd.dx0 = np.zeros(13)
    K2, _, d2 = oracle._riccati_factor(qd, np.where(cls > 0, 1e30 * np.maximum(1.0, Rd), Rd), np.zeros((N, 4)), absolute=True)
    xk = np.zeros(13); du = np.zeros((N, 4))
    for k in range(N):
        du[k] = np.where(cls[k] > 0, c[k], -K2[k] @ xk - d2[k])
        xk = qp.A[k] @ xk + qp.B[k] @ du[k]
    assert np.abs(v0 + du - ref["du"]).max() < 1e-10
