"""Mixed-horizon fleets (BASELINE.json config C5: N in {30, 50, 100}, dt fixed at 15 ms).

Thin wrapper over the C-ABI's cfnmpc_fleet_* (include/cfnmpc.h): the library buckets the vehicles
by horizon (one solver per distinct N -- a solver's workspace is blocked by stage for one
horizon), keeps the caller's vehicle order at the boundary and solves the buckets concurrently.
A bucket is also the unit of multi-GPU balancing (parallel.shard_by_horizon deals buckets out by
sum N).  Arrays may be numpy (host) or torch device tensors, as for BatchSolver."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import NU, NX, NY
from .solver import _arg, _check, default_opts, _launch_stream, _torch_device
from .synthetic import regulation_row


class MixedHorizonFleet:
    def __init__(self, horizons, **opt_kw):
        self._L = _lib.lib()
        self.horizons = np.ascontiguousarray(horizons, dtype=np.int32)
        self.B = len(self.horizons)
        self.opts = default_opts(**opt_kw)
        h = C.c_void_p()
        _check(self._L.cfnmpc_fleet_create(C.byref(h), self.B, self.horizons.ctypes.data_as(C.c_void_p), C.byref(self.opts)),
               "cfnmpc_fleet_create")
        self._h = h
        self._device = _torch_device()
        self.Nmin = self._L.cfnmpc_fleet_min_horizon(h)
        self.Nmax = self._L.cfnmpc_fleet_max_horizon(h)

    def close(self):
        if getattr(self, "_h", None):
            self._L.cfnmpc_fleet_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def workspace_bytes(self):
        return int(self._L.cfnmpc_fleet_workspace_bytes(self._h))

    def buckets(self):
        """-> [(N, fleet indices of the bucket's rows)] in ascending N"""
        out = []
        for b in range(self._L.cfnmpc_fleet_num_buckets(self._h)):
            n, c = C.c_int(0), C.c_int(0)
            _check(self._L.cfnmpc_fleet_bucket(self._h, b, C.byref(n), C.byref(c), None, None), "cfnmpc_fleet_bucket")
            idx = np.empty(c.value, dtype=np.int32)
            _check(self._L.cfnmpc_fleet_bucket(self._h, b, None, None, None, idx.ctypes.data_as(C.c_void_p)), "cfnmpc_fleet_bucket")
            out.append((n.value, idx))
        return out

    def bucket_iterates(self):
        """-> [(N, fleet indices, x [count][N+1][13], u [count][N][4])] per bucket: the persistent iterates (nlp_out of
        acados_mpc.cpp:77) through the buckets' own solvers (cfnmpc_fleet_bucket + cfnmpc_get_iterate), host arrays"""
        out = []
        for b in range(self._L.cfnmpc_fleet_num_buckets(self._h)):
            n, c, sv = C.c_int(0), C.c_int(0), C.c_void_p()
            _check(self._L.cfnmpc_fleet_bucket(self._h, b, C.byref(n), C.byref(c), C.byref(sv), None), "cfnmpc_fleet_bucket")
            idx = np.empty(c.value, dtype=np.int32)
            _check(self._L.cfnmpc_fleet_bucket(self._h, b, None, None, None, idx.ctypes.data_as(C.c_void_p)), "cfnmpc_fleet_bucket")
            x = np.empty((c.value, n.value + 1, NX)); u = np.empty((c.value, n.value, NU))
            _check(self._L.cfnmpc_get_iterate(sv, x.ctypes.data_as(C.c_void_p), u.ctypes.data_as(C.c_void_p), 0,
                                              _launch_stream(None, self._device)), "cfnmpc_get_iterate")
            out.append((n.value, idx, x, u))
        return out

    def set_regulation(self, xyz, uss):
        """xyz [B][3]: Regulation reference of every vehicle (acados_mpc.cpp:435-454)."""
        rows = np.stack([regulation_row(xyz[i], uss) for i in range(self.B)])
        self.set_yref(np.repeat(rows[:, None, :], self.Nmax, 1).copy(), rows[:, :13].copy())

    def set_yref(self, yref, yref_e):
        """yref [B][Nmax][17] (vehicle i uses rows 0..N_i-1), yref_e [B][13]"""
        p, dev, st, _k = _arg(yref, (self.B, self.Nmax, NY), device=self._device)
        pe, deve, _st, _k2 = _arg(yref_e, (self.B, NX), device=self._device)
        if dev != deve:
            raise ValueError("yref and yref_e must live on the same side")
        _check(self._L.cfnmpc_fleet_set_yref(self._h, p, pe, dev, st), "cfnmpc_fleet_set_yref")

    def set_x0(self, x0):
        p, dev, st, _k = _arg(x0, (self.B, NX), device=self._device)
        _check(self._L.cfnmpc_fleet_set_x0(self._h, p, dev, st), "cfnmpc_fleet_set_x0")

    def set_weights(self, W=None, WN=None):
        w = None if W is None else np.ascontiguousarray(W, dtype=np.float64)
        wn = None if WN is None else np.ascontiguousarray(WN, dtype=np.float64)
        _check(self._L.cfnmpc_fleet_set_weights(self._h, None if w is None else w.ctypes.data_as(C.c_void_p),
                                                None if wn is None else wn.ctypes.data_as(C.c_void_p)), "cfnmpc_fleet_set_weights")

    def set_box(self, u_min, u_max):
        _check(self._L.cfnmpc_fleet_set_box(self._h, float(u_min), float(u_max)), "cfnmpc_fleet_set_box")

    def set_box_stages(self, lb=None, ub=None):
        """per-stage / per-input boxes, host arrays [B][Nmax][4] (vehicle i uses rows 0..N_i-1); None, None: scalar box"""
        if lb is None and ub is None:
            _check(self._L.cfnmpc_fleet_set_box_stages(self._h, None, None), "cfnmpc_fleet_set_box_stages")
            return
        lb = np.ascontiguousarray(lb, dtype=np.float64); ub = np.ascontiguousarray(ub, dtype=np.float64)
        assert lb.shape == (self.B, self.Nmax, 4) and ub.shape == lb.shape
        _check(self._L.cfnmpc_fleet_set_box_stages(self._h, lb.ctypes.data_as(C.c_void_p), ub.ctypes.data_as(C.c_void_p)), "cfnmpc_fleet_set_box_stages")

    def get_cmd(self, cmd_vel=None, motvel=None):
        """Output stage of the reference node for the whole fleet (cfnmpc_fleet_get_cmd)."""
        if cmd_vel is None:
            cmd_vel = np.empty((self.B, 4))
        if motvel is None:
            if type(cmd_vel).__module__.startswith("torch"):
                import torch
                motvel = torch.empty((self.B, 4), dtype=torch.int32, device=cmd_vel.device)
            else:
                motvel = np.empty((self.B, 4), dtype=np.int32)
        p, dev, st, _k = _arg(cmd_vel, (self.B, 4), device=self._device)
        pm, devm, _s, _k2 = _arg(motvel, (self.B, 4), np.int32, device=self._device)
        assert dev == devm
        _check(self._L.cfnmpc_fleet_get_cmd(self._h, p, pm, dev, st), "cfnmpc_fleet_get_cmd")
        return cmd_vel, motvel

    def init_iterate(self, mode, stream=None):
        _check(self._L.cfnmpc_fleet_init_iterate(self._h, int(mode), _launch_stream(stream, self._device)), "cfnmpc_fleet_init_iterate")

    def solve(self, n_rti=1, stream=None):
        _check(self._L.cfnmpc_fleet_solve(self._h, int(n_rti), _launch_stream(stream, self._device)), "cfnmpc_fleet_solve")

    def get_u(self, stage, out=None):
        if out is None:
            out = np.empty((self.B, NU))
        p, dev, st, _k = _arg(out, (self.B, NU), device=self._device)
        _check(self._L.cfnmpc_fleet_get_u(self._h, int(stage), p, dev, st), "cfnmpc_fleet_get_u")
        return out

    def get_x(self, stage, out=None):
        if out is None:
            out = np.empty((self.B, NX))
        p, dev, st, _k = _arg(out, (self.B, NX), device=self._device)
        _check(self._L.cfnmpc_fleet_get_x(self._h, int(stage), p, dev, st), "cfnmpc_fleet_get_x")
        return out

    def stats(self, out=None):
        """-> (status, qp_iter, res) [B]; `out` = three torch device tensors (int32, int32, float64)
        to keep them on the device"""
        if out is not None:
            st, it, rs = out
            ps, dev, strm, _a = _arg(st, (self.B,), np.int32, device=self._device)
            pi, _d, _s, _b = _arg(it, (self.B,), np.int32, device=self._device)
            pr, _d2, _s2, _c = _arg(rs, (self.B,), device=self._device)
            _check(self._L.cfnmpc_fleet_get_stats(self._h, ps, pi, pr, dev, strm), "cfnmpc_fleet_get_stats")
            return out
        st = np.empty(self.B, dtype=np.int32); it = np.empty(self.B, dtype=np.int32); rs = np.empty(self.B)
        _check(self._L.cfnmpc_fleet_get_stats(self._h, st.ctypes.data_as(C.c_void_p), it.ctypes.data_as(C.c_void_p),
                                              rs.ctypes.data_as(C.c_void_p), 0, _launch_stream(None, self._device)), "cfnmpc_fleet_get_stats")
        return st, it, rs
