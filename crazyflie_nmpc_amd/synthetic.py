"""Synthetic inputs for benchmarks and examples: perturbed hover states and
the Regulation reference window of the reference node (acados_mpc.cpp:435-454).
Pure input generation -- no solver arithmetic lives here."""
from __future__ import annotations

import numpy as np

# generate_c_code.py:53-58: hov_w = sqrt(mq*g0/(4*Ct)) with g0 = 9.8066, mq = 0.033, Ct = 3.25e-4
HOV_W = float(np.sqrt((33e-3 * 9.8066) / (4 * 3.25e-4)))


def sample_hover_x0(rng, n, center=(0.0, 0.0, 0.4), scale=1.0):
    """Perturbed hover states, configs C2/C3: position U(+-0.3 m), roll/pitch U(+-10 deg),
    yaw U(+-20 deg) as a unit quaternion (w >= 0), body velocity U(+-0.5 m/s), rates U(+-1 rad/s).
    State order of the reference node: p(3) q(w,x,y,z) v_body(3) w_body(3)."""
    pos = np.asarray(center) + scale * rng.uniform(-0.3, 0.3, (n, 3))
    roll = scale * np.deg2rad(rng.uniform(-10, 10, n))
    pitch = scale * np.deg2rad(rng.uniform(-10, 10, n))
    yaw = scale * np.deg2rad(rng.uniform(-20, 20, n))
    cr, sr, cp, sp, cy, sy = np.cos(roll / 2), np.sin(roll / 2), np.cos(pitch / 2), np.sin(pitch / 2), np.cos(yaw / 2), np.sin(yaw / 2)
    qw = cr * cp * cy + sr * sp * sy
    qx = sr * cp * cy - cr * sp * sy
    qy = cr * sp * cy + sr * cp * sy
    qz = cr * cp * sy - sr * sp * cy
    sgn = np.where(qw < 0, -1.0, 1.0)
    quat = np.stack([qw, qx, qy, qz], axis=1) * sgn[:, None]
    vel = scale * rng.uniform(-0.5, 0.5, (n, 3))
    rate = scale * rng.uniform(-1.0, 1.0, (n, 3))
    return np.concatenate([pos, quat, vel, rate], axis=1)


def regulation_row(xyz=(0.0, 0.0, 0.4), uss=HOV_W):
    """One row of the Regulation reference window (acados_mpc.cpp:438-454)."""
    row = np.zeros(17)
    row[0:3] = xyz
    row[3] = 1.0
    row[13:17] = uss
    return row
