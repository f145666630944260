"""Per-stage / per-input input boxes (cfnmpc_set_box_stages; acados' "lbu" / "ubu" on individual stages --
the reference's FIXED_U0 variant pins stage 0 to the input in flight, acados_mpc.cpp:605-608): the HIP path
against the dense active-set oracle on the same QP, and the acados-named drop-in driven by the node twin
with FIXED_U0 against the batch API."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOV = 15.777730167256925


def test_per_stage_boxes_match_dense_oracle(oracle):
    from crazyflie_nmpc_amd import BatchSolver, default_opts
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    B, N = 48, 50
    rng = np.random.default_rng(11)
    x0 = oracle.sample_hover_x0(rng, B, scale=1.2)
    yr, ye = oracle.regulation_yref(N, (0.0, 0.0, 0.4))
    yref = np.repeat(yr[None], B, 0).copy(); yref_e = np.repeat(ye[None], B, 0).copy()
    lb = np.zeros((B, N, 4)); ub = np.full((B, N, 4), 22.0)
    pin = rng.uniform(12.0, 19.0, (B, 4))
    lb[:, 0] = pin; ub[:, 0] = pin                      # stage 0 pinned (an equality per input)
    lb[:, 3, 2] = 13.0; ub[:, 3, 2] = 17.0              # a tighter box on one input of stage 3
    ub[:, 5:9, 1] = 18.0                                # a lower ceiling on input 1 of stages 5..8
    for ah in (0, 1):
        s = BatchSolver(B, default_opts(active_horizon=ah))
        s.set_x0(x0); s.set_yref(yref, yref_e); s.init_iterate(INIT_HOVER)
        s.set_box_stages(lb, ub)
        s.solve(1)
        st, it, _ = s.stats()
        xg, ug = s.get_iterate()
        assert (st == 0).all() and (it > 0).all() and (it <= 12).all()
        assert np.abs(ug[:, 0] - pin).max() < 1e-12                 # the pinned inputs, to rounding
        assert (ug >= lb - 1e-9).all() and (ug <= ub + 1e-9).all()
        for i in range(0, B, 5):
            xbar = np.repeat(x0[i][None], N + 1, 0); ubar = np.full((N, 4), HOV)
            qp = oracle.build_qp(xbar, ubar, x0[i], yref[i], yref_e[i])
            qp.lb = lb[i] - ubar; qp.ub = ub[i] - ubar
            ref = oracle.pdas_dense(qp, max_solves=30)
            assert ref["converged"]
            assert np.abs(ug[i] - ubar - ref["du"]).max() < 1e-8, (ah, i, np.abs(ug[i] - ubar - ref["du"]).max())
            assert np.abs(xg[i] - xbar - ref["dx"]).max() < 1e-8
        # back to the scalar box: the same results as a solver that never saw per-stage boxes
        s.set_box_stages(None, None)
        f = BatchSolver(B, default_opts(active_horizon=ah))
        for q in (s, f):
            q.set_x0(x0); q.init_iterate(INIT_HOVER)
        f.set_yref(yref, yref_e)
        s.solve(1); f.solve(1)
        xs, us = s.get_iterate(); xf, uf = f.get_iterate()
        assert np.abs(us - uf).max() < 1e-9 and np.abs(xs - xf).max() < 1e-9


def test_box_stages_validation():
    from crazyflie_nmpc_amd import BatchSolver, default_opts
    s = BatchSolver(4)
    lb = np.zeros((4, 50, 4)); ub = np.full((4, 50, 4), 22.0)
    bad = ub.copy(); bad[2, 7, 1] = -1.0
    with pytest.raises(Exception):
        s.set_box_stages(lb, bad)                      # lb > ub
    nan = lb.copy(); nan[0, 0, 0] = np.nan
    with pytest.raises(Exception):
        s.set_box_stages(nan, ub)
    s.set_box_stages(lb, ub)
    with pytest.raises(Exception):
        BatchSolver(4, default_opts(cond_N2=10)).set_box_stages(lb, ub)   # not on the condensed path


def test_dropin_fixed_u0_variant_matches_batch_api(tmp_path, oracle):
    """The node twin built like the reference with FIXED_U0 1 (lbu = ubu = u1 on stage 0 before every solve)
    through the acados-named drop-in: u0 of every step IS the pinned input, and the closed loop equals the
    batch API driven with the same per-stage boxes; spot checks against the dense oracle."""
    from crazyflie_nmpc_amd import BatchSolver, sim
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "harness"), "-s"])
    exe = os.path.join(ROOT, "tests", "harness", "cf_nmpc_replay")
    x0 = np.array([0.25, -0.2, 0.55, 1, 0, 0, 0, 0.1, -0.1, 0.05, 0, 0, 0.0])
    np.savetxt(tmp_path / "x0.txt", x0[None])
    steps = 25
    subprocess.check_call([exe, "regulation", "-", str(steps), str(tmp_path / "x0.txt"), "1", str(tmp_path / "out.csv"), repr(HOV), "1"])
    got = np.loadtxt(tmp_path / "out.csv", delimiter=",")
    assert (got[:, 1] == 0).all()
    N = 50
    yr, ye = oracle.regulation_yref(N, (0.0, 0.0, 0.4), uss=HOV)
    s = BatchSolver(1)
    s.set_x0(x0[None]); s.set_yref(yr[None].copy(), ye[None].copy()); s.init_iterate(INIT_HOVER)
    x = x0[None].copy()
    u1 = np.full(4, HOV)
    for t in range(steps):
        lb = np.zeros((1, N, 4)); ub = np.full((1, N, 4), 22.0)
        lb[0, 0] = u1; ub[0, 0] = u1
        xbar, ubar = s.get_iterate()
        s.set_box_stages(lb, ub); s.set_x0(x); s.solve(1)
        st, it, _ = s.stats()
        assert st[0] == 0
        xg, ug = s.get_iterate()
        assert np.abs(ug[0, 0] - u1).max() < 1e-12                       # pinned
        assert np.abs(got[t, 3:7] - ug[0, 0]).max() < 1e-9               # drop-in: u0
        assert np.abs(got[t, 7:11] - ug[0, 1]).max() < 1e-9              # u1
        assert np.abs(got[t, 11:24] - xg[0, 4]).max() < 1e-9             # x4
        if t in (0, 3, 11):
            qp = oracle.build_qp(xbar[0], ubar[0], x[0], yr, ye)
            qp.lb = lb[0] - ubar[0]; qp.ub = ub[0] - ubar[0]
            ref = oracle.pdas_dense(qp, max_solves=30)
            assert ref["converged"] and np.abs(ug[0] - ubar[0] - ref["du"]).max() < 1e-8
        u1 = ug[0, 1].copy()
        x = sim(x, ug[:, 0], T=0.015, steps=1)


def test_fleet_per_stage_boxes_reach_every_bucket(oracle):
    """cfnmpc_fleet_set_box_stages: a mixed-horizon fleet with stage 0 pinned per vehicle equals one BatchSolver per
    horizon with the same boxes (bitwise: the bucket IS such a solver), and u0 is the pin."""
    from crazyflie_nmpc_amd import BatchSolver, default_opts
    from crazyflie_nmpc_amd.fleet import MixedHorizonFleet
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    rng = np.random.default_rng(17)
    Bf = 45
    hz = rng.choice([30, 50, 100], size=Bf)
    x0 = oracle.sample_hover_x0(rng, Bf, scale=1.3)
    f = MixedHorizonFleet(hz)
    f.set_regulation(np.tile([0.0, 0.0, 0.4], (Bf, 1)), HOV)
    lb = np.zeros((Bf, 100, 4)); ub = np.full((Bf, 100, 4), 22.0)
    pin = rng.uniform(12.0, 19.0, (Bf, 4))
    lb[:, 0] = pin; ub[:, 0] = pin
    f.set_box_stages(lb, ub)
    f.set_x0(x0); f.init_iterate(INIT_HOVER); f.solve(1)
    st, it, _ = f.stats()
    u0 = f.get_u(0); u1 = f.get_u(1)
    assert (st == 0).all() and np.abs(u0 - pin).max() < 1e-12
    for n in (30, 50, 100):
        idx = np.nonzero(hz == n)[0]
        yr, ye = oracle.regulation_yref(int(n), (0.0, 0.0, 0.4), uss=HOV)
        s = BatchSolver(len(idx), default_opts(N=int(n)))
        s.set_yref(np.repeat(yr[None], len(idx), 0).copy(), np.repeat(ye[None], len(idx), 0).copy())
        s.set_x0(x0[idx]); s.init_iterate(INIT_HOVER)
        s.set_box_stages(lb[idx, :n].copy(), ub[idx, :n].copy())
        s.solve(1)
        assert np.array_equal(s.get_u(0), u0[idx]) and np.array_equal(s.get_u(1), u1[idx])
    f.set_box_stages(None, None)
    f.set_x0(x0); f.init_iterate(INIT_HOVER); f.solve(1)
    assert np.abs(f.get_u(0) - pin).max() > 0.1          # the pin is gone
    f.close()


def test_box_stages_device_arrays_equal_host_arrays(oracle):
    """per-instance, per-stage boxes handed over as DEVICE tensors (transposed on the device into the home 4-vector layout)
    give bitwise the host-array result on a ragged fleet"""
    import torch
    from crazyflie_nmpc_amd import BatchSolver, default_opts
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    B, N = 9, 50
    rng = np.random.default_rng(21)
    x0 = oracle.sample_hover_x0(rng, B, scale=2.0)
    yr, ye = oracle.regulation_yref(N, (0.0, 0.0, 0.4))
    lb = rng.uniform(0.0, 6.0, (B, N, 4)); ub = rng.uniform(17.0, 22.0, (B, N, 4))    # every instance and stage its own box
    out = []
    for dev_arrays in (False, True):
        s = BatchSolver(B, default_opts())
        s.set_x0(x0); s.set_yref(np.repeat(yr[None], B, 0).copy(), np.repeat(ye[None], B, 0).copy()); s.init_iterate(INIT_HOVER)
        if dev_arrays:
            s.set_box_stages(torch.from_numpy(lb).cuda(), torch.from_numpy(ub).cuda())
        else:
            s.set_box_stages(lb, ub)
        s.solve(3)
        st, it, _ = s.stats()
        out.append((st, it) + s.get_iterate())
        s.close()
    (st1, it1, x1, u1), (st2, it2, x2, u2) = out
    assert (st1 == 0).all() and np.array_equal(st1, st2) and np.array_equal(it1, it2)
    assert (it1 > 0).any()
    assert np.array_equal(u1, u2) and np.array_equal(x1, x2)
    assert (u1 >= lb - 1e-9).all() and (u1 <= ub + 1e-9).all()
