"""Python mirror of the reference node's per-step protocol (crazyflie_controller/src/
acados_mpc.cpp:427-670) for a BATCH of vehicles: reference-window state machine
(Regulation / Tracking / Position_Hold), solver call, output post-processing into the wire
units of /crazyflie/cmd_vel and /crazyflie/acados_motvel.  The C++ twin for one vehicle on the
acados-named drop-in is csrc/cf_nmpc_node.hpp."""
from __future__ import annotations

import numpy as np

from .solver import BatchSolver

REGULATION, TRACKING, POSITION_HOLD = 0, 1, 2
G0_NODE = 9.80665  # the node's constant (acados_mpc.cpp:107); the model uses 9.8066


def uss_node():
    """Steady-state propeller speed exactly as the node computes it (acados_mpc.cpp:189,247-253):
    `float uss = sqrt((mq*g0)/(4*Ct))` with float mq, Ct and the double macro g0 -- i.e. mq*g0 and
    the quotient are evaluated in double, 4*Ct in float, and the result is rounded to float."""
    mq, ct = np.float32(33e-3), np.float32(3.25e-4)
    num = float(mq) * G0_NODE
    den = float(np.float32(4.0) * ct)
    return float(np.float32(np.sqrt(num / den)))


def quatern2euler(q):
    """acados_mpc.cpp:384-404; q[..., 4] = (w, x, y, z) -> (phi, theta, psi)"""
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R11 = 2 * (w * w + x * x) - 1
    R21 = 2 * (x * y - w * z)
    R31 = 2 * (x * z + w * y)
    R32 = 2 * (y * z - w * x)
    R33 = 2 * (w * w + z * z) - 1
    return np.arctan2(R32, R33), -np.arcsin(R31), np.arctan2(R21, R11)


def krpm2pwm(krpm):
    """acados_mpc.cpp:421-425: truncation toward zero into int"""
    return np.trunc(((np.asarray(krpm) * 1000) - 4070.3) / 0.2685).astype(np.int64)


def postprocess(u0, u1, x4):
    """-> dict(motvel int32 [B][4], cmd_vel float [B][4] = pitch deg, -roll deg, thrust PWM, yaw rate deg/s)
    (acados_mpc.cpp:628-670)"""
    q = x4[:, 3:7] / np.linalg.norm(x4[:, 3:7], axis=1, keepdims=True)
    phi, theta, _psi = quatern2euler(q)
    cmd = np.stack([np.rad2deg(theta), -np.rad2deg(phi), krpm2pwm(u1.mean(axis=1)).astype(np.float64),
                    np.rad2deg(x4[:, 12])], axis=1)
    return dict(motvel=np.trunc(u0).astype(np.int32), cmd_vel=cmd)


class BatchNMPC:
    """B vehicles, each with the reference node's reference-mode state machine."""

    def __init__(self, batch, traj=None, opts=None, uss=None):
        self.B = batch
        self.solver = BatchSolver(batch, opts)
        self.N = self.solver.N
        self.traj = None if traj is None else np.asarray(traj, dtype=np.float64)
        self.n_steps = 0 if traj is None else self.traj.shape[0]
        self.uss = uss_node() if uss is None else uss
        self.policy = np.full(batch, REGULATION, dtype=np.int32)
        self.iter = np.zeros(batch, dtype=np.int64)
        self.des = np.tile(np.array([0.0, 0.0, 0.40]), (batch, 1))  # config/crazyflie_params.cfg:15-17
        self.yref_sign = np.zeros((batch, self.N + 1, 17))

    def _hold_rows(self, xyz):
        row = np.zeros((xyz.shape[0], 17))
        row[:, 0:3] = xyz
        row[:, 3] = 1.0
        row[:, 13:17] = self.uss
        return np.repeat(row[:, None, :], self.N + 1, 1)

    def windows(self):
        """Fill yref_sign per vehicle according to its policy (acados_mpc.cpp:430-516)."""
        N = self.N
        if self.n_steps < N + 1:      # no usable trajectory: Tracking / Position_Hold vehicles are served as Regulation
            self.yref_sign[:] = self._hold_rows(self.des)     # (as cfnmpc_set_yref_windows does with n_rows < N + 1)
            return self.yref_sign[:, :N, :], self.yref_sign[:, N, :13]
        reg = self.policy == REGULATION
        if reg.any():
            self.yref_sign[reg] = self._hold_rows(self.des[reg])
        trk = np.where(self.policy == TRACKING)[0]
        for i in trk:
            if self.iter[i] < self.n_steps - N:
                self.yref_sign[i] = self.traj[self.iter[i]:self.iter[i] + N + 1]
                self.iter[i] += 1
            else:
                self.policy[i] = POSITION_HOLD  # window of the previous step is kept for this one (:486)
        hold = self.policy == POSITION_HOLD
        hold[trk] = False  # vehicles that switched in this very step keep their last window
        if hold.any():
            self.yref_sign[hold] = self._hold_rows(np.tile(self.traj[self.n_steps - 1, 0:3], (int(hold.sum()), 1)))
        return self.yref_sign[:, :N, :], self.yref_sign[:, N, :13]

    def iteration(self, x_est):
        yref, yref_e = self.windows()
        self.solver.set_x0(x_est)
        self.solver.set_yref(np.ascontiguousarray(yref), np.ascontiguousarray(yref_e))
        self.solver.solve(1)
        u0, u1, x4 = self.solver.get_u(0), self.solver.get_u(1), self.solver.get_x(4)
        st, it, res = self.solver.stats()
        out = postprocess(u0, u1, x4)
        out.update(u0=u0, u1=u1, x4=x4, status=st, qp_iter=it, res=res)
        return out
