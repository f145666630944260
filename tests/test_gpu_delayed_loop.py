"""Config C5 as a CLOSED loop with the 60 ms input delay (launch/acados_predictor.launch:62, acados_mpc.cpp:624), under both
predictor protocols, HIP fleet against the CPU restatement step by step:

  queued : x0 = the state predicted through the four inputs in flight, oldest first (what bench.py's C5 workload runs);
  latest : the REFERENCE's estimator -- ONE crazyflie_acados_sim_solve over the whole delay with the latest input held
           (acados_estimator.cpp:573-593).

The HIP fleet drives the loop (plant, input queue, kicks); the restatement solves the same QP sequence bucket by bucket from
the same x0 with its iterate re-synchronised to the fleet's after every step (as tests/test_gpu_parity.py does), so one
trajectory is compared and rounding cannot branch it.  The second test lets the restatement run its OWN closed loop under the
reference's protocol: the degradation bench.py reports for it (constrained fraction -> 1, QP failures after some 40 steps) is
a property of plant + protocol, present in the restatement as well -- not of the engine."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HOV = 15.777730167256925
KICK = 20


def _setup(oracle, B, seed):
    rng = np.random.default_rng(seed)
    hz = rng.choice([30, 50, 100], size=B)
    tgt = np.concatenate([rng.uniform(-1, 1, (B, 2)), rng.uniform(0.2, 1.0, (B, 1))], axis=1)
    off = np.concatenate([tgt - [0.0, 0.0, 0.4], np.zeros((B, 10))], axis=1)
    x = oracle.sample_hover_x0(rng, B) + off
    cohort = (B + KICK - 1) // KICK
    kicks = oracle.sample_hover_x0(rng, cohort * KICK).reshape(KICK, cohort, 13)
    rows = np.stack([np.concatenate([tgt[i], [1, 0, 0, 0, 0, 0, 0, 0, 0, 0], np.full(4, HOV)]) for i in range(B)])
    return hz, tgt, off, x, cohort, kicks, rows


def _predict(simf, x, uq, t, protocol):
    if protocol == "latest":
        return simf(x, uq[(t + 3) % 4], T=0.06, steps=4)
    xp = x
    for j in range(4):
        xp = simf(xp, uq[(t + j) % 4], T=0.015, steps=1)
    return xp


@pytest.mark.parametrize("protocol", ["queued", "latest"])
def test_delayed_closed_loop_matches_restatement_step_by_step(oracle, cref, protocol):
    from crazyflie_nmpc_amd import sim
    from crazyflie_nmpc_amd.fleet import MixedHorizonFleet
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    B, STEPS = 96, 40
    hz, tgt, off, x, cohort, kicks, rows = _setup(oracle, B, 515)
    # queued: every QP is well conditioned and settled by exact active-set solves on both sides: 1e-7 on every vehicle and step.
    # latest: the reference's protocol sends vehicles tumbling (inputs at both bounds over all 100 stages); their condensed
    # Hessians reach cond(H) ~ 1e11 and THREE solvers -- the engine, the C restatement, a dense numpy interior point on the
    # condensed QP -- then differ pairwise by 1e-6 .. 3e-4 at residuals below 1e-9 (measured; tightening tol to 1e-11 changes
    # nothing).  There the comparison is: median below 1e-9, 90 % of the vehicle-steps within 1e-7, 99 % within 1e-5, and on the worst one the engine is at least as
    # close to the restatement as either is to the dense referee.
    tol = 1e-8
    fleet = MixedHorizonFleet(hz, tol=tol)
    fleet.set_regulation(tgt, HOV)
    fleet.set_x0(x); fleet.init_iterate(INIT_HOVER)
    buckets = {}
    for n, idx, _x, _u in fleet.bucket_iterates():
        buckets[n] = dict(idx=idx, opts=cref.default_opts(N=int(n), active_set=1, tol=tol),
                          yref=np.repeat(rows[idx, None, :], n, 1).copy(), yref_e=rows[idx, :13].copy(),
                          xr=np.repeat(x[idx, None, :], n + 1, 1).copy(), ur=np.full((len(idx), n, 4), HOV))
    uq = [np.full((B, 4), HOV) for _ in range(4)]
    compared = constrained = unequal_solves = 0
    errs, worst, worst_case = [], 0.0, None
    for t in range(STEPS):
        c0 = (t % KICK) * cohort
        c1 = min(c0 + cohort, B)
        if c1 > c0:      # a kicked vehicle is a fresh one: new state, hover inputs in flight
            x[c0:c1] = kicks[t % KICK, : c1 - c0] + off[c0:c1]
            for q in uq:
                q[c0:c1] = HOV
        xp = _predict(sim, x, uq, t, protocol)
        fleet.set_x0(xp); fleet.solve(1)
        st, it, _ = fleet.stats()
        u0 = fleet.get_u(0)
        its = fleet.bucket_iterates()
        for n, idx, xg, ug in its:
            b = buckets[n]
            xbar, ubar = b["xr"].copy(), b["ur"].copy()
            st_r, it_r, _, _ = cref.rti_step(b["opts"], b["xr"], b["ur"], xp[idx].copy(), b["yref"], b["yref_e"], nthreads=0)
            assert np.array_equal(st[idx], st_r), (protocol, t, n, st[idx], st_r)              # same statuses, vehicle by vehicle
            assert np.array_equal(it[idx] > 0, it_r > 0), (protocol, t, n)
            ok = (st_r == 0)
            same = ok & (it[idx] == it_r)
            unequal_solves += int((ok & ~same).sum())
            # exact active-set solutions on both sides: FP64-level agreement wherever the solve counts coincide; a row that
            # fell back to the interior point (tol 1e-8) on either side agrees at the level of that tolerance
            e_row = np.abs(ug - b["ur"]).reshape(len(idx), -1).max(1)
            err = e_row[same].max(initial=0.0)
            errs.extend(e_row[same].tolist())
            if protocol == "queued":
                assert err < 1e-7, (protocol, t, n, err)
                assert np.abs(xg[same] - b["xr"][same]).max(initial=0.0) < 1e-7
                assert np.abs(u0[idx][same] - b["ur"][same][:, 0]).max(initial=0.0) < 1e-7
            assert np.abs(ug[ok] - b["ur"][ok]).max(initial=0.0) < 1e-3, (protocol, t, n)
            if err > worst:
                r = int(np.argmax(np.where(same, e_row, -1.0)))
                worst, worst_case = err, (xbar[r], ubar[r], xp[idx][r].copy(), b["yref"][r], b["yref_e"][r], ug[r].copy(), b["ur"][r].copy())
            compared += int(same.sum()); constrained += int((it_r > 0).sum())
            b["xr"][:] = xg; b["ur"][:] = ug      # one trajectory: the restatement continues from the fleet's iterate
        xn = sim(x, uq[t % 4], T=0.015, steps=1)       # the plant sees the input computed 4 periods ago
        uq[t % 4] = u0.copy()
        x = xn
    assert compared > 0.97 * B * STEPS and unequal_solves < 0.03 * B * STEPS, (compared, unequal_solves)
    assert constrained > B            # the constrained QP path was exercised throughout
    errs = np.sort(np.array(errs))
    if protocol == "queued":
        assert np.isfinite(x).all() and np.abs(x[:, :3] - tgt).max() < 1.0     # ... and this loop regulates
    else:
        qs = [float(errs[int(f * (len(errs) - 1))]) for f in (0.5, 0.9, 0.95, 0.99, 1.0)]
        assert qs[0] < 1e-9 and qs[1] < 1e-7 and qs[3] < 1e-5 and qs[4] < 1e-3, qs
        if worst > 1e-7:      # the worst vehicle-step against an independent dense solve of the same QP (numpy, sympy Jacobians)
            xb, ub, x0w, yr, ye, u_hip, u_ref = worst_case
            qp = oracle.build_qp(xb, ub, x0w, yr, ye)
            u_dense = ub + oracle.solve_qp_dense(qp, tol=1e-12, max_iter=200)["du"]
            d_hip, d_ref = np.abs(u_hip - u_dense).max(), np.abs(u_ref - u_dense).max()
            assert worst <= 2.0 * max(d_hip, d_ref), (worst, d_hip, d_ref)
            assert d_hip < 2e-3
    fleet.close()


def test_reference_predictor_protocol_degrades_in_the_restatement_too(oracle, cref):
    """bench.py reports that under the reference's own predictor (latest input held over 60 ms) this plant -- raw motor speeds,
    no onboard attitude loop -- leaves the healthy regime after some 40 steps.  Here both sides run their OWN closed loop under
    that protocol from the same start: the constrained fraction climbs the same way and neither stays healthy -- plant +
    protocol, not the engine.  Under the queued protocol both stay healthy over the same span."""
    from crazyflie_nmpc_amd import sim
    from crazyflie_nmpc_amd.fleet import MixedHorizonFleet
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    B, STEPS = 96, 80

    def run(side, protocol):
        hz, tgt, off, x, cohort, kicks, rows = _setup(oracle, B, 616)
        simf = sim if side == "hip" else cref.sim
        if side == "hip":
            fleet = MixedHorizonFleet(hz)
            fleet.set_regulation(tgt, HOV); fleet.set_x0(x); fleet.init_iterate(INIT_HOVER)
        else:
            bk = {int(n): dict(idx=np.nonzero(hz == n)[0]) for n in (30, 50, 100)}
            for n, b in bk.items():
                idx = b["idx"]
                b.update(opts=cref.default_opts(N=n, active_set=1), yref=np.repeat(rows[idx, None, :], n, 1).copy(),
                         yref_e=rows[idx, :13].copy(), xr=np.repeat(x[idx, None, :], n + 1, 1).copy(), ur=np.full((len(idx), n, 4), HOV))
        uq = [np.full((B, 4), HOV) for _ in range(4)]
        frac_c, frac_ok = [], []
        for t in range(STEPS):
            c0 = (t % KICK) * cohort
            c1 = min(c0 + cohort, B)
            if c1 > c0:
                x[c0:c1] = kicks[t % KICK, : c1 - c0] + off[c0:c1]
                for q in uq:
                    q[c0:c1] = HOV
            xp = _predict(simf, x, uq, t, protocol)
            if side == "hip":
                fleet.set_x0(xp); fleet.solve(1)
                st, it, _ = fleet.stats(); u0 = fleet.get_u(0)
            else:
                st = np.empty(B, dtype=np.int32); it = np.empty(B, dtype=np.int32); u0 = np.empty((B, 4))
                for n, b in bk.items():
                    s_, i_, _, _ = cref.rti_step(b["opts"], b["xr"], b["ur"], xp[b["idx"]].copy(), b["yref"], b["yref_e"], nthreads=0)
                    st[b["idx"]] = s_; it[b["idx"]] = i_; u0[b["idx"]] = b["ur"][:, 0]
            frac_c.append(float((it > 0).mean())); frac_ok.append(float((st == 0).mean()))
            xn = simf(x, uq[t % 4], T=0.015, steps=1)
            uq[t % 4] = u0.copy()
            x = np.where(np.isfinite(xn), xn, x)      # (a vehicle whose state blew up stays where it was: both sides alike)
        if side == "hip":
            fleet.close()
        return np.array(frac_c), np.array(frac_ok), x

    for protocol in ("latest", "queued"):
        ch, okh, xh = run("hip", protocol)
        cr, okr, xr = run("cref", protocol)
        if protocol == "queued":
            assert okh.min() == 1.0 and okr.min() == 1.0 and ch[20:].mean() < 0.5 and cr[20:].mean() < 0.5
            assert np.abs(xh - xr).max() < 1e-5          # two independent loops, exact QP solutions: still the same trajectory
        else:
            # the reference's protocol: nearly every vehicle constrained on BOTH sides, and neither loop stays healthy
            assert ch[30:].mean() > 0.85 and cr[30:].mean() > 0.85, (ch[30:].mean(), cr[30:].mean())
            assert abs(ch[30:].mean() - cr[30:].mean()) < 0.1
            assert okh[40:].min() < 0.99 and okr[40:].min() < 0.99, (okh[40:].min(), okr[40:].min())
