"""Reference trajectories for the Tracking policy of the reference node.

* `load_traj_text`: the 17-column whitespace text format of crazyflie_controller/traj/*.txt
  (one row per 15 ms: x y z qw qx qy qz vbx vby vbz wx wy wz w1 w2 w3 w4), loader semantics of
  NMPC::readDataFromFile (acados_mpc.cpp:354-382): one row per line, row count = line count.
* `Figure8`: piecewise 7th-order polynomial trajectory in the crazyflie_demo CSV format
  (crazyflie_demo/scripts/figure8.csv; evaluation semantics of uav_trajectory.py:15-20 Horner,
  :97-105 piece lookup by cumulative duration).
* `figure8_reference`: SURVEY.md App. C synthesis of a 17-column NMPC reference from it
  (the reference repo has no figure-8 in crazyflie_controller/traj/, finding F6).
"""
from __future__ import annotations

import numpy as np

from .synthetic import HOV_W

TS = 0.015
USS_FILE = 15.7777  # hover speed as printed in the reference's trajectory files (helix_traj.txt:1)


def load_traj_text(path):
    rows = []
    with open(path) as f:
        for line in f:
            rows.append([float(t) for t in line.split()])
    return np.array(rows, dtype=np.float64)


def save_traj_text(path, traj):
    np.savetxt(path, traj, fmt="%.4f")


class Figure8:
    """rows: [duration, x^0..x^7, y^0..y^7, z^0..z^7, yaw^0..yaw^7] (ascending powers)."""

    def __init__(self, table):
        self.table = np.asarray(table, dtype=np.float64)
        assert self.table.ndim == 2 and self.table.shape[1] == 33
        self.duration = float(self.table[:, 0].sum())

    @staticmethod
    def _horner(p, t):
        x = 0.0
        for i in range(len(p)):
            x = x * t + p[len(p) - 1 - i]
        return x

    def eval(self, t):
        """-> (pos[3], yaw) at time t in [0, duration]"""
        assert 0.0 <= t <= self.duration + 1e-12
        cur = 0.0
        for row in self.table:
            if t < cur + row[0]:
                tt = t - cur
                return (np.array([self._horner(row[1:9], tt), self._horner(row[9:17], tt), self._horner(row[17:25], tt)]),
                        self._horner(row[25:33], tt))
            cur += row[0]
        row = self.table[-1]
        tt = row[0]
        return (np.array([self._horner(row[1:9], tt), self._horner(row[9:17], tt), self._horner(row[17:25], tt)]),
                self._horner(row[25:33], tt))


def figure8_reference(fig8: Figure8, z0=0.5, N=50, uss=USS_FILE):
    """Kinematic 17-column reference in the style of helix_traj.txt: positions
    sampled every 15 ms, identity attitude, zero velocities, hover speeds; N+1 copies of the last
    sample appended so that the Tracking window logic (rows iter..iter+N) can run to the end."""
    n = int(np.floor(fig8.duration / TS)) + 1
    rows = []
    for k in range(n):
        pos, _yaw = fig8.eval(min(TS * k, fig8.duration))
        rows.append([pos[0], pos[1], pos[2] + z0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, uss, uss, uss, uss])
    rows += [rows[-1]] * (N + 1)
    return np.array(rows, dtype=np.float64)


def tracking_window(traj, it, N=50):
    """Rows it..it+N of the trajectory as (yref [N][17], yref_e [13]) (acados_mpc.cpp:460-485)."""
    win = traj[it:it + N + 1]
    return win[:N].copy(), win[N, :13].copy()


def regulation_window(xyz, N=50, uss=HOV_W):
    row = np.zeros(17)
    row[0:3] = xyz
    row[3] = 1.0
    row[13:17] = uss
    return np.tile(row, (N, 1)), row[:13].copy()
