"""Mixed-horizon fleets (BASELINE.json config C5: N in {30, 50, 100}, dt fixed at 15 ms).

A solver object has one horizon (its workspace is wave-blocked by stage), so a mixed fleet is
bucketed by N: one BatchSolver per horizon, instances addressed through index lists.  This is
also the unit of multi-GPU balancing (parallel.shard_by_horizon deals buckets out by sum N)."""
from __future__ import annotations

import numpy as np

from .solver import BatchSolver, default_opts
from .synthetic import regulation_row


class MixedHorizonFleet:
    def __init__(self, horizons, **opt_kw):
        self.horizons = np.asarray(horizons, dtype=np.int64)
        self.B = len(self.horizons)
        self.buckets = {}
        for N in sorted(set(self.horizons.tolist())):
            idx = np.where(self.horizons == N)[0]
            self.buckets[N] = (idx, BatchSolver(len(idx), default_opts(N=int(N), **opt_kw)))

    def set_regulation(self, xyz, uss):
        """xyz [B][3]: Regulation reference of every vehicle (acados_mpc.cpp:435-454)."""
        for N, (idx, s) in self.buckets.items():
            rows = np.stack([regulation_row(xyz[i], uss) for i in idx])
            s.set_yref(np.repeat(rows[:, None, :], N, 1).copy(), rows[:, :13].copy())

    def set_x0(self, x0):
        for _N, (idx, s) in self.buckets.items():
            s.set_x0(np.ascontiguousarray(x0[idx]))

    def init_iterate(self, mode):
        for _N, (_idx, s) in self.buckets.items():
            s.init_iterate(mode)

    def solve(self, n_rti=1):
        for _N, (_idx, s) in self.buckets.items():
            s.solve(n_rti)

    def _gather(self, fn, width):
        out = np.empty((self.B, width))
        for _N, (idx, s) in self.buckets.items():
            out[idx] = fn(s)
        return out

    def get_u(self, stage):
        return self._gather(lambda s: s.get_u(stage), 4)

    def get_x(self, stage):
        return self._gather(lambda s: s.get_x(stage), 13)

    def stats(self):
        st = np.empty(self.B, dtype=np.int32); it = np.empty(self.B, dtype=np.int32); rs = np.empty(self.B)
        for _N, (idx, s) in self.buckets.items():
            a, b, c = s.stats()
            st[idx], it[idx], rs[idx] = a, b, c
        return st, it, rs
