"""Generates the committed golden fixtures under tests/golden/ (run in the build container, where
/root/reference is mounted; the GPU box only sees the .npz files).

  traj.npz          the reference's own data files, as arrays (crazyflie_controller/traj/*.txt,
                    crazyflie_demo/scripts/figure8.csv) -- inputs of the Tracking configs and the
                    one fixture of the reference that pins the dynamics
  model.npz         f, df/d[x,u] (sympy), RK4 map and its sensitivities at sampled points
  qp.npz            RTI QPs (hover perturbations, saturating cases) with their exact solutions from
                    the dense oracle solver (independent of the Riccati interior-point method)
  closed_loop.npz   u0/u1/x4 sequences of closed-loop runs (regulation, smooth_step and helix
                    tracking) with exact QP solutions at every step
  postproc.npz      quaternion -> Euler / kRPM -> PWM vectors of the node's output stage
  figure8.npz       (`python make_golden.py fig8`) config C4: positions of the figure-8 sampled every
                    15 ms BY THE REFERENCE'S OWN EVALUATOR (crazyflie_demo/scripts/uav_trajectory.py,
                    imported here from /root/reference), the 17-column reference synthesised from
                    them, and a 40-step closed loop tracking it with exact QP solutions
                    (x, u0, u1, x4, the node's cmd_vel / motvel); plus cmd_vel / motvel of the three
                    older closed loops

The oracle is the numpy/sympy restatement oracle/cfnmpc_oracle.py (parity with acados itself is
UNPINNED: see its header)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import cfnmpc_oracle as o  # noqa: E402

REF = "/root/reference"


def gen_figure8():
    """Config C4 fixtures (separate file and seed: the older fixtures stay byte-identical)."""
    sys.path.insert(0, os.path.join(REF, "crazyflie_demo/scripts"))
    import uav_trajectory                      # the REFERENCE's evaluator, run in place
    tr = uav_trajectory.Trajectory()
    tr.loadcsv(os.path.join(REF, "crazyflie_demo/scripts/figure8.csv"))
    n = int(np.floor(tr.duration / 0.015)) + 1
    pos = np.array([tr.eval(0.015 * k).pos for k in range(n)])   # 0.015 (n - 1) < duration
    table = np.loadtxt(os.path.join(REF, "crazyflie_demo/scripts/figure8.csv"), delimiter=",", skiprows=1, usecols=range(33))
    N = 50
    ref = o.figure8_rows(table, z0=0.5, N=N)
    assert ref.shape[0] == n + N + 1 and np.abs(ref[:n, :2] - pos[:, :2]).max() < 1e-14 and np.abs(ref[:n, 2] - 0.5 - pos[:, 2]).max() < 1e-14
    rng = np.random.default_rng(20200104)
    it0 = 137
    x0 = ref[it0, :13].copy()
    x0 += 0.3 * (o.sample_hover_x0(rng, 1, center=(0, 0, 0))[0] - np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0.0]))
    x0[3:7] /= np.linalg.norm(x0[3:7])
    s = o.RTISolver(init="hover", qp_solver="dense")
    x = x0.copy()
    XS, U0, U1, X4, CMD, MV = [], [], [], [], [], []
    for t in range(40):
        yref, yref_e = o.tracking_yref(ref, it0 + t, N)
        r = s.step(x, yref, yref_e)
        mv, cmd = o.node_outputs(r["u0"], r["u1"], r["x4"])
        XS.append(x.copy()); U0.append(r["u0"]); U1.append(r["u1"]); X4.append(r["x4"]); CMD.append(cmd); MV.append(mv)
        x = o.rk4(x, r["u0"])
    out = dict(pos=pos, ref=ref, iter0=it0, f8_x=np.array(XS), f8_u0=np.array(U0), f8_u1=np.array(U1), f8_x4=np.array(X4),
               f8_cmd=np.array(CMD), f8_motvel=np.array(MV))
    c = np.load(os.path.join(HERE, "closed_loop.npz"))
    for key in ("reg", "ss", "hx"):
        both = [o.node_outputs(a, b, d) for a, b, d in zip(c[key + "_u0"], c[key + "_u1"], c[key + "_x4"])]
        out[key + "_motvel"] = np.array([m for m, _ in both]); out[key + "_cmd"] = np.array([cc for _, cc in both])
    np.savez_compressed(os.path.join(HERE, "figure8.npz"), **out)
    print("figure8.npz written:", {k: np.asarray(v).shape for k, v in out.items()})


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "fig8":
        return gen_figure8()
    rng = np.random.default_rng(20200101)
    # ---- reference data files
    smooth = np.loadtxt(os.path.join(REF, "crazyflie_controller/traj/smooth_step.txt"))
    helix = np.loadtxt(os.path.join(REF, "crazyflie_controller/traj/helix_traj.txt"))
    fig8 = np.loadtxt(os.path.join(REF, "crazyflie_demo/scripts/figure8.csv"), delimiter=",", skiprows=1, usecols=range(33))
    np.savez_compressed(os.path.join(HERE, "traj.npz"), smooth_step=smooth, helix=helix, figure8=fig8)
    # the figure-8 coefficient table is also package data (bench.py --workload figure8, config C4)
    np.save(os.path.join(ROOT, "crazyflie_nmpc_amd", "data", "figure8_coeffs.npy"), fig8)

    # ---- model vectors
    pts_x = [smooth[k, :13] for k in range(0, 451, 25)]
    pts_u = [smooth[k, 13:] for k in range(0, 451, 25)]
    xr = o.sample_hover_x0(rng, 13, scale=2.0)
    for i in range(13):
        pts_x.append(xr[i]); pts_u.append(rng.uniform(0.0, 22.0, 4))
    X = np.array(pts_x); U = np.array(pts_u)
    F = np.array([o.f_expl(x, u) for x, u in zip(X, U)])
    J = np.array([o.jac_sympy(x, u) for x, u in zip(X, U)])
    PHI, A, B = zip(*[o.rk4_sens(x, u) for x, u in zip(X, U)])
    np.savez_compressed(os.path.join(HERE, "model.npz"), x=X, u=U, f=F, jac=J, phi=np.array(PHI), A=np.array(A), B=np.array(B),
                        pred_T=0.06, pred=np.array([o.predict(x, u) for x, u in zip(X, U)]))

    # ---- QP cases
    N = 50
    yr, ye = o.regulation_yref(N, (0.0, 0.0, 0.4))
    cases = []
    x0s = np.concatenate([o.sample_hover_x0(rng, 3, scale=0.3), o.sample_hover_x0(rng, 5, scale=2.0)])
    for x0 in x0s:
        xbar = np.repeat(x0[None], N + 1, 0); ubar = np.full((N, 4), o.HOV_W)
        qp = o.build_qp(xbar, ubar, x0, yr, ye)
        sol = o.solve_qp_dense(qp)
        kkt = o.kkt_residual(qp, sol["dx"], sol["du"], sol["lam_l"], sol["lam_u"])
        assert kkt["max"] < 1e-9, kkt
        cases.append((x0, sol["dx"], sol["du"], int(((sol["lam_l"] > 1e-7) | (sol["lam_u"] > 1e-7)).sum())))
    np.savez_compressed(os.path.join(HERE, "qp.npz"), x0=np.array([c[0] for c in cases]), dx=np.array([c[1] for c in cases]),
                        du=np.array([c[2] for c in cases]), n_active=np.array([c[3] for c in cases]), yref=yr, yref_e=ye)

    # ---- closed loops with exact QP solutions
    def run(x0, steps, window, init):
        s = o.RTISolver(init=init, qp_solver="dense")
        x = x0.copy()
        U0, U1, X4, XS = [], [], [], []
        for t in range(steps):
            yref, yref_e = window(t)
            r = s.step(x, yref, yref_e)
            U0.append(r["u0"]); U1.append(r["u1"]); X4.append(r["x4"]); XS.append(x.copy())
            x = o.rk4(x, r["u0"])
        return np.array(XS), np.array(U0), np.array(U1), np.array(X4)

    x0_reg = o.sample_hover_x0(rng, 1, scale=1.0)[0]
    reg = run(x0_reg, 20, lambda t: (yr, ye), "hover")
    x0_ss = smooth[0, :13].copy()
    ss = run(x0_ss, 60, lambda t: o.tracking_yref(smooth, t, N), "hover")
    x0_hx = helix[0, :13].copy(); x0_hx[0] += 0.05
    hx = run(x0_hx, 40, lambda t: o.tracking_yref(helix, t, N), "hover")
    np.savez_compressed(os.path.join(HERE, "closed_loop.npz"),
                        reg_x=reg[0], reg_u0=reg[1], reg_u1=reg[2], reg_x4=reg[3],
                        ss_x=ss[0], ss_u0=ss[1], ss_u1=ss[2], ss_x4=ss[3],
                        hx_x=hx[0], hx_u0=hx[1], hx_u1=hx[2], hx_x4=hx[3])

    # ---- output stage (acados_mpc.cpp:384-404, 421-425, 645-668)
    q = rng.standard_normal((64, 4)); q[:, 0] = np.abs(q[:, 0]) + 1.0
    qn = q / np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = qn.T
    phi = np.arctan2(2 * (y * z - w * x), 2 * (w * w + z * z) - 1)
    theta = -np.arcsin(2 * (x * z + w * y))
    psi = np.arctan2(2 * (x * y - w * z), 2 * (w * w + x * x) - 1)
    krpm = rng.uniform(4.2, 22.0, 64)
    pwm = np.array([int(((k * 1000) - 4070.3) / 0.2685) for k in krpm])
    np.savez_compressed(os.path.join(HERE, "postproc.npz"), quat=q, phi=phi, theta=theta, psi=psi, krpm=krpm, pwm=pwm)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()


# hard_qps.npz (round 3) is NOT produced by this script: its inputs (iterate, x0) are 16 QPs CAPTURED from the HIP engine's
# closed loop at twice the bench's disturbance level (vehicles that left the region of attraction: the unconstrained minimiser
# lies 100 - 6000 kRPM outside the box; tools/r3_capture_st2.py on a GPU box), its expected values are the
# numpy oracle's riccati_ipm on them (infeasible start with clip_viol = 0, clipped start with the defaults; objective of the
# condensed QP at the clipped-start solution).
