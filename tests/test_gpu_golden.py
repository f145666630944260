"""GPU suite against the committed golden fixtures (tests/golden/*.npz) and through the
acados-named drop-in: HIP path == exact QP solutions of the dense oracle (generated in the
build container by tests/golden/make_golden.py)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
HOV = 15.777730167256925


def test_sim_and_linearisation_match_golden_model_vectors():
    from crazyflie_nmpc_amd import BatchSolver, sim
    m = np.load(os.path.join(G, "model.npz"))
    n = m["x"].shape[0]
    assert np.abs(sim(m["x"], m["u"], T=0.015, steps=1) - m["phi"]).max() < 1e-13
    assert np.abs(sim(m["x"], m["u"], T=float(m["pred_T"]), steps=4) - m["pred"]).max() < 1e-13
    N = 50
    s = BatchSolver(n)
    xit = np.repeat(m["x"][:, None, :], N + 1, 1).copy(); uit = np.repeat(m["u"][:, None, :], N, 1).copy()
    row = np.zeros(17); row[3] = 1
    s.set_x0(m["x"]); s.set_yref(np.tile(row, (n, N, 1)), np.tile(row[:13], (n, 1))); s.set_iterate(xit, uit)
    s.linearise_only()
    for _rep in (0,):
        A, B, b = s.get_linearisation()
        for k in (0, 17, 49):
            assert np.abs(A[:, k] - m["A"]).max() < 1e-13       # sympy-Jacobian sensitivities
            assert np.abs(B[:, k] - m["B"]).max() < 1e-13
            assert np.abs(b[:, k] - (m["phi"] - m["x"])).max() < 1e-13


@pytest.mark.parametrize("active_set,tol,bound", [(0, 1e-12, 5e-6), (0, 1e-8, 5e-4), (1, 1e-8, 5e-9)])
def test_qp_steps_match_golden_exact_solutions(active_set, tol, bound):
    """Interior point: central-path error ~ sqrt(tol); active-set solves (the default): exact, the
    bound is the accuracy of the committed solutions themselves."""
    from crazyflie_nmpc_amd import BatchSolver, default_opts
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    q = np.load(os.path.join(G, "qp.npz"))
    n, N = q["x0"].shape[0], 50
    for ah in (0, 1):
        s = BatchSolver(n, default_opts(tol=tol, active_horizon=ah, active_set=active_set))
        s.set_x0(q["x0"]); s.set_yref(np.tile(q["yref"], (n, 1, 1)), np.tile(q["yref_e"], (n, 1))); s.init_iterate(INIT_HOVER)
        s.solve(1)
        st, it, _ = s.stats()
        xg, ug = s.get_iterate()
        assert (st == 0).all() and ((it > 0) == (q["n_active"] > 0)).all()
        assert np.abs(ug - HOV - q["du"]).max() < bound
        assert np.abs(xg - q["x0"][:, None, :] - q["dx"]).max() < bound


def _cmd_close(cmd, motvel, g_cmd, g_motvel, u0, u1):
    """cmd_vel / motvel against committed values computed from the GOLDEN u0 / u1 / x4: angles follow the
    1e-5 state tolerance; the truncated integers may differ only where the golden input sits within
    that tolerance of an integer boundary."""
    assert np.abs(cmd[[0, 1, 3]] - g_cmd[[0, 1, 3]]).max() < 2e-3          # degrees
    assert abs(cmd[2] - g_cmd[2]) <= 1.0                                  # PWM counts (1e-5 kRPM = 0.04 counts)
    near = np.abs(u0 - np.round(u0)) < 1e-4
    assert (motvel[~near] == g_motvel[~near]).all()


def test_closed_loops_match_golden_through_batch_node():
    """Regulation + Tracking (smooth_step, helix, and config C4's figure-8) through the Python mirror
    of NMPC::iteration, with the device output stage (cfnmpc_get_cmd) beside it."""
    from crazyflie_nmpc_amd import default_opts, sim
    from crazyflie_nmpc_amd.node import BatchNMPC, TRACKING, postprocess
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    c = dict(np.load(os.path.join(G, "closed_loop.npz")))
    f8 = np.load(os.path.join(G, "figure8.npz"))
    c.update({k: f8[k] for k in f8.files})
    t = np.load(os.path.join(G, "traj.npz"))
    for key, traj, steps, it0 in (("reg", None, 20, 0), ("ss", t["smooth_step"], 60, 0), ("hx", t["helix"], 40, 0),
                                  ("f8", f8["ref"], 40, int(f8["iter0"]))):
        nm = BatchNMPC(1, traj=traj, opts=default_opts(tol=1e-12), uss=HOV)
        if traj is not None:
            nm.policy[:] = TRACKING
            nm.iter[:] = it0
        x = c[key + "_x"][0:1].copy()
        nm.solver.set_x0(x); nm.solver.init_iterate(INIT_HOVER)
        for k in range(steps):
            assert np.abs(x - c[key + "_x"][k]).max() < 1e-5, (key, k)
            out = nm.iteration(x)
            assert out["status"][0] == 0
            assert np.abs(out["u0"][0] - c[key + "_u0"][k]).max() < 1e-5, (key, k)   # kRPM
            assert np.abs(out["u1"][0] - c[key + "_u1"][k]).max() < 1e-5, (key, k)
            assert np.abs(out["x4"][0] - c[key + "_x4"][k]).max() < 1e-5, (key, k)
            # output stage on the device: identical to the host mirror on the same iterate, close to
            # the committed wire values
            cmd, mv = nm.solver.get_cmd()
            assert (mv == out["motvel"]).all() and cmd[0, 2] == out["cmd_vel"][0, 2], (key, k)
            assert np.abs(cmd - out["cmd_vel"]).max() < 1e-11, (key, k)
            _cmd_close(cmd[0], mv[0], c[key + "_cmd"][k], c[key + "_motvel"][k], c[key + "_u0"][k], c[key + "_u1"][k])
            x = sim(x, out["u0"], T=0.015, steps=1)


def test_device_output_stage_matches_golden_vectors():
    """cfnmpc_get_cmd (k_postproc) against tests/golden/postproc.npz: bit-exact on the integers (PWM,
    motor speeds), 1e-12 on the angles; host and device pointers; fleet variant."""
    import torch
    from crazyflie_nmpc_amd import BatchSolver
    p = np.load(os.path.join(G, "postproc.npz"))
    B, N = 64, 50
    s = BatchSolver(B)
    x = np.zeros((B, N + 1, 13)); x[:, :, 3] = 1.0
    u = np.full((B, N, 4), HOV)
    x[:, 4, 3:7] = p["quat"]                       # unnormalised on purpose: the node normalises x4's quaternion
    x[:, 4, 12] = np.linspace(-3.0, 3.0, B)
    u[:, 1, :] = p["krpm"][:, None]                # four equal speeds: their mean is the value itself
    u[:, 0, :] = np.linspace(0.2, 21.9, B * 4).reshape(B, 4)
    s.set_iterate(x, u)
    cmd, mv = s.get_cmd()
    assert (cmd[:, 2] == p["pwm"]).all()                                       # (int)((krpm*1000-4070.3)/0.2685)
    assert (mv == np.trunc(u[:, 0, :]).astype(np.int32)).all() and mv.dtype == np.int32
    assert np.abs(cmd[:, 0] - np.rad2deg(p["theta"])).max() < 1e-12            # pitch  = +deg(theta)
    assert np.abs(cmd[:, 1] + np.rad2deg(p["phi"])).max() < 1e-12              # roll   = -deg(phi)
    assert np.abs(cmd[:, 3] - np.rad2deg(x[:, 4, 12])).max() < 1e-12
    dev = torch.device("cuda", 0)
    cmd_d = torch.empty((B, 4), dtype=torch.float64, device=dev)
    cmd_d, mv_d = s.get_cmd(cmd_d)
    torch.cuda.synchronize()
    assert np.array_equal(cmd_d.cpu().numpy(), cmd) and np.array_equal(mv_d.cpu().numpy(), mv)


def test_acados_dropin_replay_matches_goldens(tmp_path):
    """The ROS-free C++ twin of the reference node (tests/harness/cf_nmpc_node.hpp), which defines the
    acados globals itself and calls acados_create/solve/ocp_nlp_* exactly like acados_mpc.cpp,
    reproduces the COMMITTED closed loops (closed_loop.npz: exact QP solutions of the dense oracle) and
    their wire values (figure8.npz), and agrees with the batch API to rounding."""
    from crazyflie_nmpc_amd import default_opts, sim
    from crazyflie_nmpc_amd.node import BatchNMPC, TRACKING
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    from crazyflie_nmpc_amd.trajectories import save_traj_text
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "harness"), "-s"])
    exe = os.path.join(ROOT, "tests", "harness", "cf_nmpc_replay")
    t = np.load(os.path.join(G, "traj.npz"))
    c = np.load(os.path.join(G, "closed_loop.npz"))
    f8 = np.load(os.path.join(G, "figure8.npz"))
    for mode, key, steps in (("regulation", "reg", 20), ("tracking", "ss", 60)):
        x0 = c[key + "_x"][0]
        np.savetxt(tmp_path / "x0.txt", x0[None], fmt="%.17g")
        trajfile = "-"
        traj = None
        if mode == "tracking":
            trajfile = str(tmp_path / "traj.txt")
            save_traj_text(trajfile, t["smooth_step"])      # the file format has 4 decimals, as the data
            traj = np.loadtxt(trajfile)
            assert np.array_equal(traj, t["smooth_step"])
        out_csv = tmp_path / f"{mode}.csv"
        # hold rows with the MODEL's hover speed, as the goldens were generated
        run = subprocess.run([exe, mode, trajfile, str(steps), str(tmp_path / "x0.txt"), "1", str(out_csv), repr(HOV)],
                             check=True, capture_output=True, text=True)
        # single-instance latency (config C1): acados_solve() must fit the node's 15 ms period
        lat = [ln for ln in run.stderr.splitlines() if "acados_solve() wall time" in ln]
        assert lat, run.stderr
        assert float(lat[0].split("median")[1].split()[0]) < 15.0, lat[0]
        R = np.loadtxt(out_csv, delimiter=",")
        assert R.shape == (steps, 3 + 4 + 4 + 13 + 4 + 4 + 2)
        assert (R[:, 1] == 0).all()                                  # acados_solve() status
        nm = BatchNMPC(1, traj=traj, opts=default_opts(), uss=HOV)
        if mode == "tracking":
            nm.policy[:] = TRACKING
        x = x0[None].copy()
        nm.solver.set_x0(x); nm.solver.init_iterate(INIT_HOVER)
        for k in range(steps):
            u0, u1, x4 = R[k, 3:7], R[k, 7:11], R[k, 11:24]
            # (1) the committed closed loop (default tolerance 1e-8, exact active-set solves)
            assert np.abs(u0 - c[key + "_u0"][k]).max() < 1e-5 and np.abs(u1 - c[key + "_u1"][k]).max() < 1e-5, (key, k)
            assert np.abs(x4 - c[key + "_x4"][k]).max() < 1e-5, (key, k)
            _cmd_close(R[k, 24:28], R[k, 28:32], f8[key + "_cmd"][k], f8[key + "_motvel"][k], c[key + "_u0"][k], c[key + "_u1"][k])
            # (2) the batch API on the same inputs
            out = nm.iteration(x)
            assert np.abs(out["u0"][0] - u0).max() < 1e-9 and np.abs(out["u1"][0] - u1).max() < 1e-9
            assert np.abs(out["x4"][0] - x4).max() < 1e-9
            assert int(R[k, 33]) == int(out["qp_iter"][0])
            x = sim(x, out["u0"], T=0.015, steps=1)


def test_figure8_tracking_config4_runs_and_tracks():
    """Config C4: figure-8 reference synthesised per SURVEY App. C, per-instance phase offsets."""
    from crazyflie_nmpc_amd import sim
    from crazyflie_nmpc_amd.node import BatchNMPC, TRACKING
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    from crazyflie_nmpc_amd.synthetic import sample_hover_x0
    from crazyflie_nmpc_amd.trajectories import Figure8, figure8_reference
    t = np.load(os.path.join(G, "traj.npz"))
    ref = figure8_reference(Figure8(t["figure8"]), z0=0.5)
    B = 64
    rng = np.random.default_rng(20200104)
    nm = BatchNMPC(B, traj=ref)
    nm.policy[:] = TRACKING
    nm.iter[:] = rng.integers(0, 436, B)
    x = ref[nm.iter, :13].copy()
    x += 0.3 * (sample_hover_x0(rng, B, center=(0, 0, 0)) - np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0.0]))
    x[:, 3:7] /= np.linalg.norm(x[:, 3:7], axis=1, keepdims=True)
    nm.solver.set_x0(x); nm.solver.init_iterate(INIT_HOVER)
    for k in range(40):
        out = nm.iteration(x)
        assert (out["status"] == 0).all()
        x = sim(x, out["u0"], T=0.015, steps=1)
    err = np.abs(x[:, :3] - ref[nm.iter, :3]).max()
    assert err < 0.15, err                       # metres: the fleet follows the figure-8


def test_device_reference_windows_match_node_state_machine():
    """cfnmpc_set_yref_windows (device) == BatchNMPC.windows() (host mirror of
    acados_mpc.cpp:430-516) for a fleet mixing Regulation / Tracking / Position_Hold, including the
    Tracking -> Position_Hold hand-over, and gives the same controls."""
    import torch
    from crazyflie_nmpc_amd import BatchSolver, sim
    from crazyflie_nmpc_amd.node import BatchNMPC, TRACKING, REGULATION
    from crazyflie_nmpc_amd.solver import INIT_HOVER
    from crazyflie_nmpc_amd.synthetic import sample_hover_x0
    t = np.load(os.path.join(G, "traj.npz"))
    traj = t["helix"]
    B = 23
    rng = np.random.default_rng(9)
    dev = torch.device("cuda", 0)
    nm = BatchNMPC(B, traj=traj)
    nm.policy[:] = np.where(np.arange(B) % 3 == 0, REGULATION, TRACKING)
    nm.iter[:] = rng.integers(0, 1000, B)
    nm.iter[1] = 998; nm.iter[2] = 999; nm.iter[4] = 1000      # around the end: N_STEPS - N = 1000
    nm.des[:] = rng.uniform(-1, 1, (B, 3)) * [1, 1, 0.4] + [0, 0, 0.6]
    x = sample_hover_x0(rng, B)
    x[:, :3] += np.where((nm.policy == TRACKING)[:, None], traj[np.minimum(nm.iter, 1049), :3] - [0, 0, 0.4], nm.des - [0, 0, 0.4])
    s = BatchSolver(B)
    mode = torch.from_numpy(nm.policy.astype(np.int32)).to(dev)
    it = torch.from_numpy(nm.iter.astype(np.int32)).to(dev)
    des = torch.from_numpy(nm.des.copy()).to(dev)
    trj = torch.from_numpy(traj.copy()).to(dev)
    nm.solver.set_x0(x); nm.solver.init_iterate(INIT_HOVER)
    s.set_x0(x); s.init_iterate(INIT_HOVER)
    for k in range(4):
        out = nm.iteration(x)                                 # host windows + solve
        s.set_yref_windows(trj, mode, it, des, nm.uss)       # device windows
        s.set_x0(x); s.solve(1)
        torch.cuda.synchronize()
        assert np.array_equal(mode.cpu().numpy(), nm.policy) and np.array_equal(it.cpu().numpy(), nm.iter), k
        assert np.array_equal(s.get_u(0), out["u0"]) and np.array_equal(s.get_x(4), out["x4"]), k
        x = sim(x, out["u0"], T=0.015, steps=1)
    assert (nm.policy == 2).sum() >= 3                        # the hand-over to Position_Hold happened


def test_estimator_state_assembly_and_predictor_match_oracle(oracle):
    """cfnmpc_estimate == numpy restatement of ESTIMATOR::predictor over several timer ticks
    (filter state carried), for both velocity branches."""
    import torch
    from crazyflie_nmpc_amd import estimate
    rng = np.random.default_rng(31)
    B = 300
    dev = torch.device("cuda", 0)
    for use_lpf in (True, False):
        filt = np.concatenate([rng.uniform(-1, 1, (B, 3)), rng.uniform(-0.5, 0.5, (B, 6))], axis=1)
        filt_d = torch.from_numpy(filt.copy()).to(dev)
        pos = filt[:, :3].copy()
        for tick in range(4):
            pos = pos + rng.uniform(-0.01, 0.01, (B, 3))
            meas = np.concatenate([pos, rng.uniform(-25, 25, (B, 3)), rng.uniform(-2, 2, (B, 3))], axis=1)
            if tick == 0:
                meas[0, 3:6] = [170.0, -40.0, 175.0]          # exercises the w < 0 sign flip
            u = rng.uniform(5, 20, (B, 4))
            xe, xp = estimate(torch.from_numpy(meas).to(dev), filt_d, torch.from_numpy(u).to(dev), dt=0.015,
                              use_lpf=use_lpf, delay=0.06, steps=4)
            xe, xp = xe.cpu().numpy(), xp.cpu().numpy()
            for i in range(0, B, 7):
                we, wp = oracle.estimator_step(meas[i], filt[i], u[i], 0.015, use_lpf, 0.06, 4)
                assert np.abs(xe[i] - we).max() < 1e-12 and np.abs(xp[i] - wp).max() < 1e-12
                assert abs(np.linalg.norm(xe[i, 3:7]) - 1) < 1e-14 and xe[i, 3] >= 0
            ok = np.arange(0, B, 7)
            assert np.abs(filt_d.cpu().numpy()[ok] - filt[ok]).max() < 1e-13
            filt = filt_d.cpu().numpy().copy()               # keep host copy in sync for untested rows
