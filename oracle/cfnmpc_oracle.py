"""CPU oracle (numpy, FP64) for the Crazyflie SQP-RTI hot path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference's arithmetic for this path lives in acados / HPIPM / BLASFEO
(github.com/tmmsartor/acados, branch `crazyflie`, no pinned commit, empty submodule under
/root/reference/acados; .gitmodules:7-10) and in the git-ignored generated solver
(`c_generated_code/`).  Neither can be built or imported here, and the reference ships no
golden vectors or tests for the path.  This file restates the *mathematics* the reference
pins:

  * ODE + constants ............ crazyflie_controller/scripts/crazyflie_full_model/export_ode_model.py:33-102
  * OCP (N, Tf, W, bounds) ..... crazyflie_controller/scripts/crazyflie_full_model/generate_c_code.py:41-146
  * per-step host protocol ..... crazyflie_controller/src/acados_mpc.cpp:427-670
  * predictor protocol ......... crazyflie_controller/src/acados_estimator.cpp:521-593

and is pinned only by (G1) traj/smooth_step.txt (RK4, dt=0.015 reproduces row k+1 from row k),
(G2/G3) the hover fixed point / RTI known answer and (G4/G5) file constants, see
tests/test_oracle_golden.py.  The QP produced by one RTI step is strictly convex, so its
solution is unique: two independent solvers live here (a dense primal-dual solver on the
condensed QP, `solve_qp_dense`, and the stage-wise Riccati interior point method that the HIP
kernels implement, `riccati_ipm`) and are cross-checked through the KKT conditions; `solve_qp_refined`
(round 6) is the REFEREE between FP64 solvers that disagree above their tolerances: the same QP solved in
x87 extended precision with refined linear solves and an extended-precision KKT check.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import numpy as np

# ----------------------------------------------------------------------------------------
# constants  (export_ode_model.py:34-42, generate_c_code.py:41-59)
# ----------------------------------------------------------------------------------------
G0 = 9.8066
MQ = 33e-3
IXX = 1.395e-5
IYY = 1.395e-5
IZZ = 2.173e-5
CD = 7.9379e-06
CT = 3.25e-4
DQ = 65e-3
ARM = DQ / 2

NX, NU, NY, NYN = 13, 4, 17, 13
N_DEFAULT = 50
TF_DEFAULT = 0.75
DT = TF_DEFAULT / N_DEFAULT  # 0.015
HOV_W = float(np.sqrt((MQ * G0) / (4 * CT)))  # generate_c_code.py:58
U_MIN, U_MAX = 0.0, 22.0  # generate_c_code.py:133-134

# generate_c_code.py:63-84,109
W_DIAG = np.array([120.0, 100.0, 100.0, 1e-3, 1e-3, 1e-3, 1e-3, 0.7, 1.0, 4.0, 1e-5, 1e-5, 10.0,
                   0.06, 0.06, 0.06, 0.06])
Q_DIAG = W_DIAG[:NX].copy()
R_DIAG = W_DIAG[NX:].copy()
WN_FACTOR = 50.0
QN_DIAG = WN_FACTOR * Q_DIAG


# ----------------------------------------------------------------------------------------
# dynamics  (export_ode_model.py:85-101); x = [p(3) q(4: w x y z) vb(3) w(3)], u = krpm(4)
# ----------------------------------------------------------------------------------------
def f_expl(x, u):
    """Continuous-time ODE right-hand side; works on (..., 13), (..., 4) arrays."""
    x = np.asarray(x, dtype=np.float64)
    u = np.asarray(u, dtype=np.float64)
    q1, q2, q3, q4 = x[..., 3], x[..., 4], x[..., 5], x[..., 6]
    vbx, vby, vbz = x[..., 7], x[..., 8], x[..., 9]
    wx, wy, wz = x[..., 10], x[..., 11], x[..., 12]
    w1, w2, w3, w4 = u[..., 0], u[..., 1], u[..., 2], u[..., 3]
    out = np.empty(np.broadcast(x[..., 0], u[..., 0]).shape + (NX,))
    out[..., 0] = vbx * (2 * q1**2 + 2 * q2**2 - 1) - vby * (2 * q1 * q4 - 2 * q2 * q3) + vbz * (2 * q1 * q3 + 2 * q2 * q4)
    out[..., 1] = vby * (2 * q1**2 + 2 * q3**2 - 1) + vbx * (2 * q1 * q4 + 2 * q2 * q3) - vbz * (2 * q1 * q2 - 2 * q3 * q4)
    out[..., 2] = vbz * (2 * q1**2 + 2 * q4**2 - 1) - vbx * (2 * q1 * q3 - 2 * q2 * q4) + vby * (2 * q1 * q2 + 2 * q3 * q4)
    out[..., 3] = -(q2 * wx) / 2 - (q3 * wy) / 2 - (q4 * wz) / 2
    out[..., 4] = (q1 * wx) / 2 - (q4 * wy) / 2 + (q3 * wz) / 2
    out[..., 5] = (q4 * wx) / 2 + (q1 * wy) / 2 - (q2 * wz) / 2
    out[..., 6] = (q2 * wy) / 2 - (q3 * wx) / 2 + (q1 * wz) / 2
    out[..., 7] = vby * wz - vbz * wy + G0 * (2 * q1 * q3 - 2 * q2 * q4)
    out[..., 8] = vbz * wx - vbx * wz - G0 * (2 * q1 * q2 + 2 * q3 * q4)
    out[..., 9] = vbx * wy - vby * wx - G0 * (2 * q1**2 + 2 * q4**2 - 1) + (CT * (w1**2 + w2**2 + w3**2 + w4**2)) / MQ
    out[..., 10] = -(CT * ARM * (w1**2 + w2**2 - w3**2 - w4**2) - IYY * wy * wz + IZZ * wy * wz) / IXX
    out[..., 11] = -(CT * ARM * (w1**2 - w2**2 - w3**2 + w4**2) + IXX * wx * wz - IZZ * wx * wz) / IYY
    out[..., 12] = -(CD * (w1**2 - w2**2 + w3**2 - w4**2) - IXX * wx * wy + IYY * wx * wy) / IZZ
    return out


_SYM_CACHE = {}


def sympy_model():
    """Symbolic restatement (sympy) used to derive the exact Jacobian independently of the
    hand-written one in the C/HIP code.  Returns (f_lambda, J_lambda) taking 17 scalars."""
    if "f" in _SYM_CACHE:
        return _SYM_CACHE["f"], _SYM_CACHE["J"]
    import sympy as sp

    xs = sp.symbols("xq yq zq q1 q2 q3 q4 vbx vby vbz wx wy wz")
    us = sp.symbols("w1 w2 w3 w4")
    xq, yq, zq, q1, q2, q3, q4, vbx, vby, vbz, wx, wy, wz = xs
    w1, w2, w3, w4 = us
    g0, mq, Ixx, Iyy, Izz, Cd, Ct, l = G0, MQ, IXX, IYY, IZZ, CD, CT, ARM
    fx = sp.Matrix([
        vbx * (2 * q1**2 + 2 * q2**2 - 1) - vby * (2 * q1 * q4 - 2 * q2 * q3) + vbz * (2 * q1 * q3 + 2 * q2 * q4),
        vby * (2 * q1**2 + 2 * q3**2 - 1) + vbx * (2 * q1 * q4 + 2 * q2 * q3) - vbz * (2 * q1 * q2 - 2 * q3 * q4),
        vbz * (2 * q1**2 + 2 * q4**2 - 1) - vbx * (2 * q1 * q3 - 2 * q2 * q4) + vby * (2 * q1 * q2 + 2 * q3 * q4),
        -(q2 * wx) / 2 - (q3 * wy) / 2 - (q4 * wz) / 2,
        (q1 * wx) / 2 - (q4 * wy) / 2 + (q3 * wz) / 2,
        (q4 * wx) / 2 + (q1 * wy) / 2 - (q2 * wz) / 2,
        (q2 * wy) / 2 - (q3 * wx) / 2 + (q1 * wz) / 2,
        vby * wz - vbz * wy + g0 * (2 * q1 * q3 - 2 * q2 * q4),
        vbz * wx - vbx * wz - g0 * (2 * q1 * q2 + 2 * q3 * q4),
        vbx * wy - vby * wx - g0 * (2 * q1**2 + 2 * q4**2 - 1) + (Ct * (w1**2 + w2**2 + w3**2 + w4**2)) / mq,
        -(Ct * l * (w1**2 + w2**2 - w3**2 - w4**2) - Iyy * wy * wz + Izz * wy * wz) / Ixx,
        -(Ct * l * (w1**2 - w2**2 - w3**2 + w4**2) + Ixx * wx * wz - Izz * wx * wz) / Iyy,
        -(Cd * (w1**2 - w2**2 + w3**2 - w4**2) - Ixx * wx * wy + Iyy * wx * wy) / Izz,
    ])
    allv = list(xs) + list(us)
    J = fx.jacobian(allv)
    _SYM_CACHE["f"] = sp.lambdify(allv, fx, "numpy")
    _SYM_CACHE["J"] = sp.lambdify(allv, J, "numpy")
    _SYM_CACHE["Jsym"] = J
    return _SYM_CACHE["f"], _SYM_CACHE["J"]


def jac_sympy(x, u):
    """13x17 Jacobian [df/dx | df/du] from the sympy model (single point)."""
    _, J = sympy_model()
    return np.asarray(J(*list(x), *list(u)), dtype=np.float64).reshape(NX, NX + NU)


def jac_fd(x, u, eps=1e-6):
    """Central finite-difference Jacobian (sanity check only)."""
    z = np.concatenate([x, u])
    J = np.zeros((NX, NX + NU))
    for i in range(NX + NU):
        zp, zm = z.copy(), z.copy()
        zp[i] += eps
        zm[i] -= eps
        J[:, i] = (f_expl(zp[:NX], zp[NX:]) - f_expl(zm[:NX], zm[NX:])) / (2 * eps)
    return J


# ----------------------------------------------------------------------------------------
# integrator: classic RK4, one step per shooting interval, with forward
# variational equations (the role of acados sim_erk + CasADi forw_vde, acados_mpc.cpp:84)
# ----------------------------------------------------------------------------------------
def rk4(x, u, dt=DT, steps=1):
    x = np.asarray(x, dtype=np.float64)
    h = dt / steps
    for _ in range(steps):
        k1 = f_expl(x, u)
        k2 = f_expl(x + 0.5 * h * k1, u)
        k3 = f_expl(x + 0.5 * h * k2, u)
        k4 = f_expl(x + h * k3, u)
        x = x + (h / 6.0) * (k1 + 2 * k2 + 2 * k3 + k4)
    return x


def rk4_sens(x, u, dt=DT, jac=jac_sympy):
    """One RK4 step with forward sensitivities.  Returns (Phi, A[13x13], B[13x4])."""
    x = np.asarray(x, dtype=np.float64)
    u = np.asarray(u, dtype=np.float64)
    S0 = np.hstack([np.eye(NX), np.zeros((NX, NU))])  # d x / d [x0 u]

    def stage(xs, Ss):
        J = jac(xs, u)
        k = f_expl(xs, u)
        K = J[:, :NX] @ Ss
        K[:, NX:] += J[:, NX:]
        return k, K

    k1, K1 = stage(x, S0)
    k2, K2 = stage(x + 0.5 * dt * k1, S0 + 0.5 * dt * K1)
    k3, K3 = stage(x + 0.5 * dt * k2, S0 + 0.5 * dt * K2)
    k4, K4 = stage(x + dt * k3, S0 + dt * K3)
    phi = x + (dt / 6.0) * (k1 + 2 * k2 + 2 * k3 + k4)
    S = S0 + (dt / 6.0) * (K1 + 2 * K2 + 2 * K3 + K4)
    return phi, S[:, :NX].copy(), S[:, NX:].copy()


def predict(x, u, delay=0.06, steps=4):
    """Delay-compensating predictor (acados_estimator.cpp:573-593): RK4 over `delay`
    with `steps` sub-steps (4 steps of 0.015)."""
    return rk4(x, u, dt=delay, steps=steps)


# ----------------------------------------------------------------------------------------
# Gauss-Newton QP of one RTI step (generate_c_code.py:62-146, acados_mpc.cpp:581-594)
# ----------------------------------------------------------------------------------------
class StageQP:
    """Stage-wise data of the RTI QP in step variables (dx, du).

        min  sum_k 1/2 dx'Q dx + q_k'dx + 1/2 du'R du + r_k'du  + 1/2 dx_N'QN dx_N + q_N'dx_N
        s.t. dx_0 = x0 - xbar_0 ;  dx_{k+1} = A_k dx_k + B_k du_k + b_k
             lb_k <= du_k <= ub_k        (lb_k = U_MIN - ubar_k, ub_k = U_MAX - ubar_k)
    """

    def __init__(self, N):
        self.N = N
        self.A = np.zeros((N, NX, NX))
        self.B = np.zeros((N, NX, NU))
        self.b = np.zeros((N, NX))
        self.q = np.zeros((N + 1, NX))
        self.r = np.zeros((N, NU))
        self.lb = np.zeros((N, NU))
        self.ub = np.zeros((N, NU))
        self.dx0 = np.zeros(NX)
        self.Qd = Q_DIAG.copy()
        self.Rd = R_DIAG.copy()
        self.QNd = QN_DIAG.copy()


def build_qp(xbar, ubar, x0, yref, yref_e, dt=DT, jac=jac_sympy,
             u_min=U_MIN, u_max=U_MAX):
    """xbar (N+1,13), ubar (N,4): current iterate; yref (N,17); yref_e (13)."""
    N = ubar.shape[0]
    qp = StageQP(N)
    for k in range(N):
        phi, A, B = rk4_sens(xbar[k], ubar[k], dt, jac)
        qp.A[k], qp.B[k] = A, B
        qp.b[k] = phi - xbar[k + 1]
        qp.q[k] = Q_DIAG * (xbar[k] - yref[k, :NX])
        qp.r[k] = R_DIAG * (ubar[k] - yref[k, NX:])
        qp.lb[k] = u_min - ubar[k]
        qp.ub[k] = u_max - ubar[k]
    qp.q[N] = QN_DIAG * (xbar[N] - yref_e)
    qp.dx0 = x0 - xbar[0]
    return qp


def condense(qp: StageQP):
    """Full condensing: dx = G du + g  ->  dense QP in du (4N variables)."""
    N = qp.N
    nU = N * NU
    Gam = np.zeros(((N + 1) * NX, nU))
    g = np.zeros((N + 1) * NX)
    g[:NX] = qp.dx0
    for k in range(N):
        r0, r1 = k * NX, (k + 1) * NX
        Gam[r1:r1 + NX, :] = qp.A[k] @ Gam[r0:r1, :]
        Gam[r1:r1 + NX, k * NU:(k + 1) * NU] += qp.B[k]
        g[r1:r1 + NX] = qp.A[k] @ g[r0:r1] + qp.b[k]
    Qbar = np.concatenate([np.tile(qp.Qd, N), qp.QNd])
    qbar = qp.q.reshape(-1)
    Rbar = np.tile(qp.Rd, N)
    H = (Gam.T * Qbar) @ Gam + np.diag(Rbar)
    h = Gam.T @ (Qbar * g + qbar) + qp.r.reshape(-1)
    return H, h, Gam, g


# ----------------------------------------------------------------------------------------
# Partial condensing (HPIPM d_part_cond, selected by PARTIAL_CONDENSING_HPIPM at
# generate_c_code.py:140; README.md:77).  acados / HPIPM are not under /root/reference, so this
# restates the published algorithm: the N-stage QP is regrouped into N2 blocks of consecutive
# stages; inside a block the interior states are eliminated exactly, leaving a QP with N2 stages,
# nx = 13 states and nu2 = 4 x (block length) inputs whose solution is the SAME primal point.
# ----------------------------------------------------------------------------------------
def block_sizes(N, N2):
    """Stages per block: N2 blocks, the first N mod N2 one stage longer (even split otherwise)."""
    m, rem = divmod(N, N2)
    return [m + 1] * rem + [m] * (N2 - rem)


def partial_condense(qp: StageQP, N2):
    """-> list of blocks, each dict(k0, m, D, H):  with z = (U, dx_{k0}, 1), U = (du_{k0} .. du_{k0+m-1}),
         dx_{k0+m} = D z                         D = [Bbar (13 x 4m) | Abar (13 x 13) | bbar (13)]
         cost of stages k0 .. k0+m-1 = 1/2 z'H z  H = [[Rbar, Sbar, rbar], [Sbar', Qbar, qbar], [rbar', qbar', *]]
    (forward recursion G_{i+1} = A G_i + [B at du_{k0+i}] [b at 1], H += G_i' Q~ G_i)."""
    blocks = []
    k0 = 0
    for m in block_sizes(qp.N, N2):
        mu = NU * m
        w = mu + NX + 1
        G = np.zeros((NX, w))
        G[:, mu:mu + NX] = np.eye(NX)
        H = np.zeros((w, w))
        for i in range(m):
            k = k0 + i
            Gq = G.copy()
            H += Gq.T @ (qp.Qd[:, None] * Gq)                      # 1/2 dx' Q dx
            H[:, w - 1] += Gq.T @ qp.q[k]                            # q_k' dx
            H[w - 1, :] += Gq.T @ qp.q[k]
            sl = slice(NU * i, NU * (i + 1))
            H[sl, sl] += np.diag(qp.Rd)                              # 1/2 du' R du
            H[sl, w - 1] += qp.r[k]                                  # r_k' du
            H[w - 1, sl] += qp.r[k]
            G = qp.A[k] @ G
            G[:, sl] += qp.B[k]
            G[:, w - 1] += qp.b[k]
        H[w - 1, w - 1] = 0.0
        blocks.append(dict(k0=k0, m=m, D=G, H=H))
        k0 += m
    return blocks


def riccati_condensed(blocks, QNd, qN, dx0, diag_add=None):
    """Riccati recursion over the condensed stages (one mu x mu Cholesky per block).  diag_add:
    optional list of per-block vectors added to the diagonal of Rbar (interior-point barrier).
    -> (U per block, dx at the block starts incl. the terminal state)"""
    P = np.diag(QNd).astype(np.float64)
    p = np.asarray(qN, dtype=np.float64).copy()
    Ks, ds = [], []
    for j in range(len(blocks) - 1, -1, -1):
        bk = blocks[j]
        mu = NU * bk["m"]
        w = mu + NX + 1
        E = np.vstack([bk["D"], np.eye(w)[w - 1]])                  # (14 x w): [dx+; 1] = E z
        Pt = np.zeros((NX + 1, NX + 1))
        Pt[:NX, :NX] = P; Pt[:NX, NX] = p; Pt[NX, :NX] = p
        Hh = bk["H"] + E.T @ Pt @ E
        if diag_add is not None:
            Hh[np.arange(mu), np.arange(mu)] += diag_add[j]
        L = np.linalg.cholesky(Hh[:mu, :mu])
        KD = np.linalg.solve(L.T, np.linalg.solve(L, Hh[:mu, mu:]))  # [K | d]: U = -K dx - d
        Ks.append(KD[:, :NX]); ds.append(KD[:, NX])
        Sch = Hh[mu:, mu:] - Hh[mu:, :mu] @ KD
        P = 0.5 * (Sch[:NX, :NX] + Sch[:NX, :NX].T)
        p = Sch[:NX, NX].copy()
    Ks.reverse(); ds.reverse()
    xs = [np.asarray(dx0, dtype=np.float64)]
    Us = []
    for j, bk in enumerate(blocks):
        mu = NU * bk["m"]
        U = -Ks[j] @ xs[-1] - ds[j]
        Us.append(U)
        xs.append(bk["D"] @ np.concatenate([U, xs[-1], [1.0]]))
    return Us, np.array(xs)


def expand_condensed(qp: StageQP, Us):
    """Recover every stage's (dx, du) from the condensed inputs: roll the ORIGINAL stage dynamics."""
    du = np.concatenate(Us).reshape(qp.N, NU)
    dx = np.zeros((qp.N + 1, NX))
    dx[0] = qp.dx0
    for k in range(qp.N):
        dx[k + 1] = qp.A[k] @ dx[k] + qp.B[k] @ du[k] + qp.b[k]
    return dx, du


def solve_qp_pcond(qp: StageQP, N2):
    """Unconstrained minimiser through partial condensing (pcond -> condensed Riccati -> expand)."""
    blocks = partial_condense(qp, N2)
    Us, _ = riccati_condensed(blocks, qp.QNd, qp.q[qp.N], qp.dx0)
    return expand_condensed(qp, Us)


def solve_qp_dense(qp: StageQP, tol=1e-11, max_iter=100):
    """Independent high-accuracy solver: primal-dual interior point on the condensed dense QP
    (numpy Cholesky).  Returns dict(dx, du, lam_l, lam_u, iters)."""
    H, h, Gam, g = condense(qp)
    n = H.shape[0]
    lb = qp.lb.reshape(-1)
    ub = qp.ub.reshape(-1)
    v = np.linalg.solve(H, -h)
    if np.all(v > lb) and np.all(v < ub):
        lam_l = np.zeros(n)
        lam_u = np.zeros(n)
        it = 0
    else:
        v = np.clip(v, lb + 0.05 * (ub - lb), ub - 0.05 * (ub - lb))
        tl, tu = v - lb, ub - v
        lam_l = np.ones(n)
        lam_u = np.ones(n)
        for it in range(1, max_iter + 1):
            rg = H @ v + h - lam_l + lam_u
            mu = (lam_l @ tl + lam_u @ tu) / (2 * n)
            if max(np.abs(rg).max(), (lam_l * tl).max(), (lam_u * tu).max()) < tol or mu < 1e-15:
                break

            def newton(sig_mu, cl, cu):
                D = lam_l / tl + lam_u / tu
                rhs = -rg + (sig_mu - cl) / tl - lam_l - (sig_mu - cu) / tu + lam_u
                dv = np.linalg.solve(H + np.diag(D), rhs)
                dtl, dtu = dv, -dv
                dll = (sig_mu - cl) / tl - lam_l - lam_l * dtl / tl
                dlu = (sig_mu - cu) / tu - lam_u - lam_u * dtu / tu
                return dv, dtl, dtu, dll, dlu

            def steplen(dtl, dtu, dll, dlu):
                a = 1.0
                for z, dz in ((tl, dtl), (tu, dtu), (lam_l, dll), (lam_u, dlu)):
                    m = dz < 0
                    if m.any():
                        a = min(a, float((-z[m] / dz[m]).min()))
                return a

            dv, dtl, dtu, dll, dlu = newton(0.0, 0.0, 0.0)
            a = steplen(dtl, dtu, dll, dlu)
            mu_aff = ((lam_l + a * dll) @ (tl + a * dtl) + (lam_u + a * dlu) @ (tu + a * dtu)) / (2 * n)
            sig = (mu_aff / mu) ** 3
            dv, dtl, dtu, dll, dlu = newton(sig * mu, dll * dtl, dlu * dtu)
            a = min(1.0, 0.995 * steplen(dtl, dtu, dll, dlu))
            v = v + a * dv
            tl, tu = tl + a * dtl, tu + a * dtu
            lam_l, lam_u = lam_l + a * dll, lam_u + a * dlu
    dx = (Gam @ v + g).reshape(qp.N + 1, NX)
    return dict(dx=dx, du=v.reshape(qp.N, NU), lam_l=lam_l.reshape(qp.N, NU),
                lam_u=lam_u.reshape(qp.N, NU), iters=it)


def kkt_residual(qp: StageQP, dx, du, lam_l, lam_u):
    """Independent optimality check of a candidate QP solution: recovers the costates pi from
    the backward adjoint recursion and returns the max-norm of (stationarity in du,
    dynamics, bound violation, complementarity, negative multipliers)."""
    N = qp.N
    pi = np.zeros((N + 1, NX))
    pi[N] = qp.QNd * dx[N] + qp.q[N]
    for k in range(N - 1, 0, -1):
        pi[k] = qp.Qd * dx[k] + qp.q[k] + qp.A[k].T @ pi[k + 1]
    res_g = 0.0
    res_b = float(np.abs(dx[0] - qp.dx0).max())
    for k in range(N):
        rg = qp.Rd * du[k] + qp.r[k] + qp.B[k].T @ pi[k + 1] - lam_l[k] + lam_u[k]
        res_g = max(res_g, float(np.abs(rg).max()))
        rb = qp.A[k] @ dx[k] + qp.B[k] @ du[k] + qp.b[k] - dx[k + 1]
        res_b = max(res_b, float(np.abs(rb).max()))
    viol = max(float((qp.lb - du).max()), float((du - qp.ub).max()), 0.0)
    comp = max(float(np.abs(lam_l * (du - qp.lb)).max()), float(np.abs(lam_u * (qp.ub - du)).max()))
    neg = max(float((-lam_l).max()), float((-lam_u).max()), 0.0)
    return dict(res_g=res_g, res_b=res_b, viol=viol, comp=comp, neg=neg,
                max=max(res_g, res_b, viol, comp, neg))


# ----------------------------------------------------------------------------------------
# Stage-wise Riccati interior point (the algorithm the HIP kernels implement; plays the role
# of HPIPM d_ocp_qp_ipm_solve selected at generate_c_code.py:140).  See DESIGN.md section 4.
# ----------------------------------------------------------------------------------------
IPM_DEFAULTS = dict(tol=1e-8, max_iter=50, tau=0.995, thr0=1.0, lam0_min=1e-2, mu0_scale=0.1, clip_viol=2.0, clip_margin=0.05)


def _riccati_factor(qp, Rhat, rhat, absolute=True):
    """Backward sweep.  Rhat (N,4) diagonal input Hessian, rhat (N,4) input gradient.
    absolute=True : full affine problem (q, b, dx0 of the QP)  -> value function 1/2x'Px + p'x
    absolute=False: homogeneous problem (q = 0, b = 0, dx0 = 0), only rhat drives it.
    Returns per-stage K (N,4,13), Sinv (N,4,4), d (N,4)."""
    N = qp.N
    P = np.diag(qp.QNd)
    p = qp.q[N].copy() if absolute else np.zeros(NX)
    K = np.zeros((N, NU, NX))
    Sinv = np.zeros((N, NU, NU))
    d = np.zeros((N, NU))
    for k in range(N - 1, -1, -1):
        A, B = qp.A[k], qp.B[k]
        PA, PB = P @ A, P @ B
        S = np.diag(Rhat[k]) + B.T @ PB
        G = B.T @ PA
        hb = p + P @ qp.b[k] if absolute else p
        rho = rhat[k] + B.T @ hb
        Si = np.linalg.inv(S)
        Si = 0.5 * (Si + Si.T)
        K[k] = Si @ G
        d[k] = Si @ rho
        Sinv[k] = Si
        P = np.diag(qp.Qd) + A.T @ PA - G.T @ K[k]
        P = 0.5 * (P + P.T)
        p = A.T @ hb - K[k].T @ rho
        if absolute:
            p = p + qp.q[k]
    return K, Sinv, d


def _riccati_forward(qp, K, d, absolute=True):
    N = qp.N
    x = qp.dx0.copy() if absolute else np.zeros(NX)
    xs = np.zeros((N + 1, NX))
    vs = np.zeros((N, NU))
    xs[0] = x
    for k in range(N):
        v = -K[k] @ x - d[k]
        x = qp.A[k] @ x + qp.B[k] @ v
        if absolute:
            x = x + qp.b[k]
        vs[k] = v
        xs[k + 1] = x
    return xs, vs


def _riccati_resolve(qp, K, Sinv, g):
    """Re-use a factorisation for a new right-hand side g (N,4) on the input rows
    (homogeneous problem).  Returns d (N,4) for _riccati_forward(absolute=False)."""
    N = qp.N
    p = np.zeros(NX)
    d2 = np.zeros((N, NU))
    for k in range(N - 1, -1, -1):
        rho = g[k] + qp.B[k].T @ p
        d2[k] = Sinv[k] @ rho
        p = qp.A[k].T @ p - K[k].T @ rho
    return d2


def riccati_ipm(qp: StageQP, **opts):
    """Mehrotra predictor-corrector on the box-constrained OCP-QP with stage-wise Riccati
    solves.

    Start = unconstrained minimiser (one absolute Riccati solve); if it satisfies the bounds
    it is the solution (lam = 0) and is returned.  Otherwise slacks/multipliers are shifted
    positive and Newton steps are taken in *delta* form: the x-stationarity rows and the
    dynamics rows of the KKT system hold exactly at the start and the Newton system is linear
    in them, so their right-hand sides stay zero; the input-stationarity residual r_g scales
    by (1 - alpha) per step and is carried as data; the slack residual rho = v - lb - t is
    recomputed locally.  One factorisation + two homogeneous solves per iteration."""
    o = dict(IPM_DEFAULTS)
    o.update(opts)
    N = qp.N
    nc = 2 * NU * N
    lb, ub = qp.lb, qp.ub
    Rd = np.tile(qp.Rd, (N, 1))
    K, Sinv, d = _riccati_factor(qp, Rd, qp.r, absolute=True)
    xs, v = _riccati_forward(qp, K, d, absolute=True)
    info = dict(iters=0, status=0, factorizations=1)
    if np.all(v >= lb) and np.all(v <= ub):
        z = np.zeros_like(v)
        return dict(dx=xs, du=v, lam_l=z, lam_u=z.copy(), **info)
    viol = max(float(np.maximum(lb - v, 0).max()), float(np.maximum(v - ub, 0).max()))
    if o["clip_viol"] > 0 and viol > o["clip_viol"] * float((ub - lb).max()):
        # CLIPPED START (cfnmpc_opts.ipm_clip_viol): the unconstrained minimiser lies more than clip_viol box
        # widths outside the box (vehicles far from the iterate's trajectory: ~100 kRPM and more).  From there the
        # infeasible start above spends 30-60 iterations at tiny step lengths; instead the iteration starts INSIDE the
        # box -- v clipped to a margin of clip_margin of the width -- with multipliers that absorb the gradient of the
        # condensed QP there, g = H (v - v0) (one forward sweep for dx, one backward costate sweep), and a complementarity
        # floor scaled by it: lam_l = max(g, 0) + mu0 / t_l, lam_u = max(-g, 0) + mu0 / t_u, mu0 = max(lam0_min, mu0_scale *
        # mean(|g| min(t_l, t_u))); slacks are exact (rho = 0), r_g = g - lam_l + lam_u.  (numpy: 63 -> 22 iterations on
        # captured fall-back QPs, none at the cap; 5 -> 8 on ordinary ones -- hence the threshold.)
        w = ub - lb
        vc = np.clip(v, lb + o["clip_margin"] * w, ub - o["clip_margin"] * w)
        dvc = vc - v
        dxs = np.zeros((N + 1, NX))
        for k in range(N):
            dxs[k + 1] = qp.A[k] @ dxs[k] + qp.B[k] @ dvc[k]
        pi = qp.QNd * dxs[N]
        g = np.zeros_like(v)
        for k in range(N - 1, -1, -1):
            g[k] = Rd[k] * dvc[k] + qp.B[k].T @ pi
            pi = qp.Qd * dxs[k] + qp.A[k].T @ pi
        v = vc
        tl, tu = v - lb, ub - v
        mu0 = max(o["lam0_min"], o["mu0_scale"] * float((np.abs(g) * np.minimum(tl, tu)).sum()) / (NU * N))
        ll = np.maximum(g, 0.0) + mu0 / tl
        lu = np.maximum(-g, 0.0) + mu0 / tu
        rg = g - ll + lu
    else:
        tl = np.maximum(v - lb, o["thr0"])
        tu = np.maximum(ub - v, o["thr0"])
        mu0 = max(o["lam0_min"], o["mu0_scale"] * viol)
        ll = mu0 / tl
        lu = mu0 / tu
        rg = -ll + lu  # R v + r + B'pi = 0 at the unconstrained start
    status = 2
    it = 0
    while True:
        rl = v - lb - tl
        ru = ub - v - tu
        mu = float((ll * tl).sum() + (lu * tu).sum()) / nc
        res = max(float((ll * tl).max()), float((lu * tu).max()), float(np.abs(rg).max()),
                  float(np.abs(rl).max()), float(np.abs(ru).max()))
        if res <= o["tol"]:
            status = 0
            break
        if it >= o["max_iter"]:
            break
        it += 1
        Dl, Du = ll / tl, lu / tu
        Rhat = Rd + Dl + Du
        # predictor (sigma = 0, no second-order term)
        g_aff = rg + ll + Dl * rl - lu - Du * ru
        K, Sinv, d = _riccati_factor(qp, Rhat, g_aff, absolute=False)
        info["factorizations"] += 1
        _, dv_a = _riccati_forward(qp, K, d, absolute=False)
        dtl_a = dv_a + rl
        dtu_a = -dv_a + ru
        dll_a = -ll - Dl * dtl_a
        dlu_a = -lu - Du * dtu_a
        a_aff = _steplen(tl, tu, ll, lu, dtl_a, dtu_a, dll_a, dlu_a)
        mu_aff = float(((ll + a_aff * dll_a) * (tl + a_aff * dtl_a)).sum()
                       + ((lu + a_aff * dlu_a) * (tu + a_aff * dtu_a)).sum()) / nc
        sigma = (mu_aff / mu) ** 3
        smu = sigma * mu
        # corrector: same factorisation, right-hand side changes on the input rows only
        cl = dll_a * dtl_a
        cu = dlu_a * dtu_a
        g_cor = (cl - smu) / tl - (cu - smu) / tu
        d2 = _riccati_resolve(qp, K, Sinv, g_cor)
        _, dv_c = _riccati_forward(qp, K, d2, absolute=False)
        dv = dv_a + dv_c
        dtl = dv + rl
        dtu = -dv + ru
        dll = (smu - cl) / tl - ll - Dl * dtl
        dlu = (smu - cu) / tu - lu - Du * dtu
        a = min(1.0, o["tau"] * _steplen(tl, tu, ll, lu, dtl, dtu, dll, dlu))
        v = v + a * dv
        tl, tu = tl + a * dtl, tu + a * dtu
        ll, lu = ll + a * dll, lu + a * dlu
        rg = (1.0 - a) * rg
    info.update(iters=it, status=status)
    x = qp.dx0.copy()
    xs = np.zeros((N + 1, NX))
    xs[0] = x
    for k in range(N):
        x = qp.A[k] @ x + qp.B[k] @ v[k] + qp.b[k]
        xs[k + 1] = x
    return dict(dx=xs, du=v, lam_l=ll, lam_u=lu, **info)


def pdas_dense(qp: StageQP, max_solves=12, warm_cls=None):
    """Primal-dual active-set solve of the condensed QP (dense algebra) -- the CPU statement of the
    engine's `active_set` path (include/cfnmpc.h): classify every input from
    the unconstrained minimiser (below / above its bound = active, else free), solve the QP with
    the active inputs fixed, re-classify (a free input that leaves the box becomes active; a lower-
    active one stays while its multiplier H v + h > 0, an upper-active one while it is < 0), stop
    when the classification is stationary -- which is the KKT system of the strictly convex QP.
    Returns dict(dx, du, solves, converged, cls); solves = 0 if the unconstrained minimiser is feasible.
    warm_cls (N, 4) of 0 free / 1 lower / 2 upper (cfnmpc_opts.as_warm): the final classification of the instance's previous
    RTI step; the first solve then starts from its union with today's violations.  cls = the final classification."""
    H, h, Gam, g = condense(qp)
    n = H.shape[0]
    lb = qp.lb.reshape(-1)
    ub = qp.ub.reshape(-1)
    v0 = np.linalg.solve(H, -h)
    v = v0
    eq = ~(lb < ub)                      # lb = ub (per-stage boxes, cfnmpc_set_box_stages): an equality, fixed whatever
    lo, up = (v0 < lb) | eq, (v0 > ub) & ~eq   # the sign of its multiplier
    solves, converged = 0, True
    if (lo.any() or up.any()) and warm_cls is not None:
        w = np.asarray(warm_cls).reshape(-1)
        inside = ~(lo | up)
        lo, up = lo | (inside & (w == 1)), up | (inside & (w == 2))
    if lo.any() or up.any():
        converged = False
        while solves < max_solves:
            solves += 1
            act = lo | up
            free = ~act
            v = np.where(lo, lb, np.where(up, ub, 0.0))
            if free.any():
                v[free] = np.linalg.solve(H[np.ix_(free, free)], -h[free] - H[np.ix_(free, act)] @ v[act])
            grad = H @ v + h
            lo2 = (free & (v < lb)) | (lo & (grad > 0)) | eq
            up2 = ((free & (v > ub)) | (up & (grad < 0))) & ~eq
            if np.array_equal(lo2, lo) and np.array_equal(up2, up):
                converged = True
                break
            lo, up = lo2, up2
    dx = (Gam @ v + g).reshape(qp.N + 1, NX)
    cls = np.where(lo, 1, np.where(up, 2, 0)).astype(np.uint8).reshape(qp.N, NU)
    return dict(dx=dx, du=v.reshape(qp.N, NU), solves=solves, converged=converged, cls=cls)


def qp_from_blocks(A, B, b, q, r, dx0, Qd, Rd, QNd, lb, ub):
    """StageQP from given stage blocks (e.g. the engine's or the C restatement's linearisation: cfnmpc_debug_get_linearisation,
    cref.linearise) and weights other than the generator's defaults."""
    qp = StageQP(A.shape[0])
    qp.A[:], qp.B[:], qp.b[:], qp.q[:], qp.r[:] = A, B, b, q, r
    qp.dx0 = np.asarray(dx0, dtype=np.float64).copy()
    qp.Qd, qp.Rd, qp.QNd = (np.asarray(v, dtype=np.float64).copy() for v in (Qd, Rd, QNd))
    qp.lb[:], qp.ub[:] = lb, ub
    return qp


def solve_qp_refined(qp: StageQP, max_solves=80):
    """REFEREE between two FP64 solvers that disagree above their own tolerances: the exact solution of the (strictly convex,
    hence unique: generate_c_code.py:140 names one QP, whatever solves it) RTI QP in EXTENDED precision -- the stage data are
    taken as exact, the condensed H, h are formed in x87 80-bit arithmetic (numpy longdouble, eps 1.1e-19), the primal-dual
    active-set iteration of pdas_dense runs on them with every linear solve refined against extended-precision residuals
    (FP64 LU as the preconditioner), and the KKT conditions of the result are checked in extended precision.
    -> dict(dx, du (FP64 roundings of the extended solution), cls, solves, kkt (max KKT violation), cond (of H_FF), eps)."""
    LD = np.longdouble
    N = qp.N
    nU = N * NU
    A, B, b = qp.A.astype(LD), qp.B.astype(LD), qp.b.astype(LD)
    Gam = np.zeros(((N + 1) * NX, nU), dtype=LD)
    g = np.zeros((N + 1) * NX, dtype=LD)
    g[:NX] = qp.dx0
    for k in range(N):
        r0, r1 = k * NX, (k + 1) * NX
        Gam[r1:r1 + NX, :] = A[k] @ Gam[r0:r1, :]
        Gam[r1:r1 + NX, k * NU:(k + 1) * NU] += B[k]
        g[r1:r1 + NX] = A[k] @ g[r0:r1] + b[k]
    Qbar = np.concatenate([np.tile(qp.Qd, N), qp.QNd]).astype(LD)
    H = (Gam.T * Qbar) @ Gam
    H[np.arange(nU), np.arange(nU)] += np.tile(qp.Rd, N).astype(LD)
    H = (H + H.T) / LD(2)
    h = Gam.T @ (Qbar * g + qp.q.reshape(-1).astype(LD)) + qp.r.reshape(-1).astype(LD)
    lb, ub = qp.lb.reshape(-1).astype(LD), qp.ub.reshape(-1).astype(LD)

    def solve(M, rhs):            # M x = rhs, M extended precision: FP64 factorisation + refinement on extended residuals
        import scipy.linalg as sla
        lu = sla.lu_factor(M.astype(np.float64))
        x = np.zeros(rhs.shape, dtype=LD)
        for _ in range(40):
            res = rhs - M @ x
            dxr = sla.lu_solve(lu, res.astype(np.float64)).astype(LD)
            x = x + dxr
            if np.abs(dxr).max() <= LD(4) * np.finfo(LD).eps * max(np.abs(x).max(), LD(1e-300)):
                break
        return x

    v = solve(H, -h)
    eq = ~(lb < ub)
    lo, up = (v < lb) | eq, (v > ub) & ~eq
    solves, cond = 0, float(np.linalg.cond(H.astype(np.float64)))
    while (lo.any() or up.any()) and solves < max_solves:
        solves += 1
        act = lo | up
        free = ~act
        v = np.where(lo, lb, np.where(up, ub, LD(0)))
        if free.any():
            HFF = H[np.ix_(free, free)]
            v[free] = solve(HFF, -h[free] - H[np.ix_(free, act)] @ v[act])
            cond = float(np.linalg.cond(HFF.astype(np.float64)))
        grad = H @ v + h
        lo2 = (free & (v < lb)) | (lo & (grad > 0)) | eq
        up2 = ((free & (v > ub)) | (up & (grad < 0))) & ~eq
        if np.array_equal(lo2, lo) and np.array_equal(up2, up):
            break
        lo, up = lo2, up2
    grad = H @ v + h
    free = ~(lo | up)
    kkt = max(float(np.abs(grad[free]).max()) if free.any() else 0.0,
              float(np.maximum(-grad[lo & ~eq], 0).max()) if (lo & ~eq).any() else 0.0,
              float(np.maximum(grad[up], 0).max()) if up.any() else 0.0,
              float(np.maximum(lb - v, 0).max()), float(np.maximum(v - ub, 0).max()))
    dx = (Gam @ v + g).reshape(N + 1, NX)
    cls = np.where(lo, 1, np.where(up, 2, 0)).astype(np.uint8).reshape(N, NU)
    return dict(dx=dx.astype(np.float64), du=v.reshape(N, NU).astype(np.float64), cls=cls, solves=solves, kkt=kkt, cond=cond,
                eps=float(np.finfo(LD).eps))


def _steplen(tl, tu, ll, lu, dtl, dtu, dll, dlu):
    a = 1.0
    for z, dz in ((tl, dtl), (tu, dtu), (ll, dll), (lu, dlu)):
        m = dz < 0
        if m.any():
            a = min(a, float((-z[m] / dz[m]).min()))
    return a


# ----------------------------------------------------------------------------------------
# One SQP-RTI step and the closed loop around it (acados_mpc.cpp:427-670)
# ----------------------------------------------------------------------------------------
def regulation_yref(N, xyz, uss=HOV_W):
    """Reference window of the Regulation policy (acados_mpc.cpp:435-454)."""
    row = np.zeros(NY)
    row[0:3] = xyz
    row[3] = 1.0
    row[13:17] = uss
    yref = np.tile(row, (N, 1))
    return yref, row[:NX].copy()


def tracking_yref(traj, it, N):
    """Reference window of the Tracking policy (acados_mpc.cpp:460-485): rows it..it+N."""
    win = traj[it:it + N + 1]
    return win[:N].copy(), win[N, :NX].copy()


class RTISolver:
    """SQP real-time iteration: one QP per call, full step, iterate persists, no shift
    (acados SQP_RTI as configured at generate_c_code.py:140-146; warm start = the persistent
    nlp_out of acados_mpc.cpp:77).  init='acados' is SURVEY App. D-3 (x_k = codegen x0,
    u_k = 0); init='hover' sets x_k = first x0, u_k = hover speed."""

    def __init__(self, N=N_DEFAULT, dt=DT, init="acados", qp_solver="riccati_ipm", jac=jac_sympy,
                 **ipm_opts):
        self.N, self.dt = N, dt
        self.x = np.tile(np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0.0]), (N + 1, 1))
        self.u = np.zeros((N, NU))
        self.init = init
        self.first = True
        self.qp_solver = qp_solver
        self.jac = jac
        self.ipm_opts = ipm_opts
        self.last_qp = None
        self.last_sol = None

    def step(self, x0, yref, yref_e):
        if self.first and self.init == "hover":
            self.x[:] = x0
            self.u[:] = HOV_W
        self.first = False
        qp = build_qp(self.x, self.u, np.asarray(x0, dtype=np.float64), yref, yref_e, self.dt, self.jac)
        if self.qp_solver == "riccati_ipm":
            sol = riccati_ipm(qp, **self.ipm_opts)
        else:
            sol = solve_qp_dense(qp)
            sol["status"] = 0
        self.x = self.x + sol["dx"]
        self.u = self.u + sol["du"]
        self.last_qp, self.last_sol = qp, sol
        return dict(u0=self.u[0].copy(), u1=self.u[1].copy(), x4=self.x[4].copy(),
                    status=sol["status"], iters=sol["iters"])


def sample_hover_x0(rng, n, center=(0.0, 0.0, 0.4), scale=1.0):
    """Synthetic perturbed hover states."""
    pos = np.asarray(center) + scale * rng.uniform(-0.3, 0.3, (n, 3))
    roll = scale * np.deg2rad(rng.uniform(-10, 10, n))
    pitch = scale * np.deg2rad(rng.uniform(-10, 10, n))
    yaw = scale * np.deg2rad(rng.uniform(-20, 20, n))
    cr, sr, cp, sp_, cy, sy = np.cos(roll / 2), np.sin(roll / 2), np.cos(pitch / 2), np.sin(pitch / 2), np.cos(yaw / 2), np.sin(yaw / 2)
    qw = cr * cp * cy + sr * sp_ * sy
    qx = sr * cp * cy - cr * sp_ * sy
    qy = cr * sp_ * cy + sr * cp * sy
    qz = cr * cp * sy - sr * sp_ * cy
    sgn = np.where(qw < 0, -1.0, 1.0)
    quat = np.stack([qw, qx, qy, qz], axis=1) * sgn[:, None]
    vel = scale * rng.uniform(-0.5, 0.5, (n, 3))
    rate = scale * rng.uniform(-1.0, 1.0, (n, 3))
    return np.concatenate([pos, quat, vel, rate], axis=1)



# ----------------------------------------------------------------------------------------
# figure-8 reference of config C4 and the node's output stage
# ----------------------------------------------------------------------------------------
def poly_piece_eval(table, t):
    """Position (x, y, z) of the piecewise 7th-order trajectory at time t: the piece is found by
    cumulative duration (crazyflie_demo/scripts/uav_trajectory.py:97-105), each axis evaluated by
    Horner's rule on ascending-power coefficients (:15-20).  table rows: [duration, x^0..x^7,
    y^0..y^7, z^0..z^7, yaw^0..yaw^7] (crazyflie_demo/scripts/figure8.csv)."""
    t0 = 0.0
    for row in table:
        if t < t0 + row[0]:
            break
        t0 += row[0]
    else:                       # t == total duration: end of the last piece
        t0 -= row[0]
    tt = t - t0
    out = np.zeros(3)
    for ax in range(3):
        c = row[1 + 8 * ax: 9 + 8 * ax]
        acc = 0.0
        for i in range(8):
            acc = acc * tt + c[7 - i]
        out[ax] = acc
    return out


def figure8_rows(table, z0=0.5, N=N_DEFAULT, uss=15.7777):
    """17-column kinematic NMPC reference synthesised from the polynomial table in the style of
    crazyflie_controller/traj/helix_traj.txt: one row per 15 ms, identity attitude,
    zero velocities, the files' hover speed 15.7777 (helix_traj.txt:1), N + 1 copies of the last
    sample appended for the Tracking window logic (acados_mpc.cpp:460-486)."""
    total = float(np.sum(table[:, 0]))
    n = int(np.floor(total / DT)) + 1
    rows = np.zeros((n + N + 1, NY))
    for k in range(n):
        rows[k, 0:3] = poly_piece_eval(table, min(DT * k, total))
    rows[n:, 0:3] = rows[n - 1, 0:3]
    rows[:, 2] += z0
    rows[:, 3] = 1.0
    rows[:, 13:17] = uss
    return rows


def node_outputs(u0, u1, x4):
    """What NMPC::iteration publishes (acados_mpc.cpp:628-670): motvel = u0 truncated to int32
    (msg/PropellerSpeedsStamped.msg:2-5), cmd_vel = [pitch deg, -roll deg, thrust PWM, yaw rate
    deg/s] from the normalised quaternion of x4 (:384-404), mean(u1) (:421-425) and x4.wz."""
    pi = 3.14159265358979323846       # acados_mpc.cpp:106
    q = np.asarray(x4[3:7], dtype=np.float64)
    q = q / np.sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3])
    w, x, y, z = q
    phi = np.arctan2(2 * (y * z - w * x), 2 * (w * w + z * z) - 1)
    theta = -np.arcsin(2 * (x * z + w * y))
    pwm = int((((u1[0] + u1[1] + u1[2] + u1[3]) / 4 * 1000) - 4070.3) / 0.2685)
    cmd = np.array([1.0 * (theta * 180.0 / pi), -1.0 * (phi * 180.0 / pi), float(pwm), x4[12] * 180.0 / pi])
    return np.array([int(v) for v in u0], dtype=np.int32), cmd

# ----------------------------------------------------------------------------------------
# estimator: state assembly + delay compensation (acados_estimator.cpp:327-368, 414-440, 521-634)
# ----------------------------------------------------------------------------------------
def euler2quatern(phi, theta, psi):
    """acados_estimator.cpp:327-354 (vector part negated, w >= 0)"""
    cph, sph = np.cos(phi * 0.5), np.sin(phi * 0.5)
    cth, sth = np.cos(theta * 0.5), np.sin(theta * 0.5)
    cps, sps = np.cos(psi * 0.5), np.sin(psi * 0.5)
    q = np.array([cph * cth * cps + sph * sth * sps,
                  -(cps * cth * sph - sps * sth * cph),
                  -(cps * sth * cph + sps * cth * sph),
                  -(sps * cth * cph - cps * sth * sph)])
    return -q if q[0] < 0 else q


def rotate_e2b(q, v):
    """acados_estimator.cpp:414-440"""
    w, x, y, z = q
    S = np.array([[2 * (w * w + x * x) - 1, 2 * (x * y + w * z), 2 * (x * z - w * y)],
                  [2 * (x * y - w * z), 2 * (w * w + y * y) - 1, 2 * (y * z + w * x)],
                  [2 * (x * z + w * y), 2 * (y * z - w * x), 2 * (w * w + z * z) - 1]])
    return S @ v


def estimator_step(meas, filt, u, dt=0.015, use_lpf=True, delay=0.06, steps=4):
    """One predictor() call for one vehicle.  meas = [x y z roll pitch yaw (deg, as published)
    wx wy wz]; filt = [p_prev(3), v1(3), v2(3)] is updated in place.  Returns (x_est, x_pred)."""
    phi, theta, psi = np.deg2rad(meas[3]), np.deg2rad(-meas[4]), np.deg2rad(meas[5])   # :493-496 pitch flip
    q = euler2quatern(phi, theta, psi)
    q = q / np.linalg.norm(q)                                                            # :546
    ve = np.zeros(3)
    for a in range(3):
        pk, pk1, v1, v2 = meas[a], filt[a], filt[3 + a], filt[6 + a]
        ve[a] = 0.3306 * v1 - 0.02732 * v2 + 35.7 * pk - 35.7 * pk1 if use_lpf else (pk - pk1) / dt   # :356-368
        filt[a], filt[6 + a], filt[3 + a] = pk, v1, ve[a]
    x_est = np.concatenate([meas[0:3], q, rotate_e2b(q, ve), meas[6:9]])
    return x_est, rk4(x_est, u, dt=delay, steps=steps)
