"""Batch solver object over the C-ABI (include/cfnmpc.h).

Mirrors, for B instances, the calls the reference node makes on its single generated solver
(crazyflie_controller/src/acados_mpc.cpp): acados_create (:225) -> BatchSolver(...);
lbx/ubx (:581-582) -> set_x0; yref (:590-594) -> set_yref; acados_solve (:611) -> solve;
ocp_nlp_out_get u/x (:619-625) -> get_u / get_x; status -> stats.

Arrays may be numpy (host, copied) or torch CUDA/HIP tensors (device pointers are handed to
the library, the work is enqueued on torch's current stream).  torch is plumbing only.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import NU, NX, NY, Opts

INIT_ACADOS, INIT_HOVER = 0, 1


class CfnmpcError(RuntimeError):
    pass


def _check(rc, what):
    if rc != 0:
        raise CfnmpcError(f"{what} failed with code {rc}")


def default_opts(**kw) -> Opts:
    o = Opts()
    _check(_lib.lib().cfnmpc_default_opts_v(C.byref(o), C.sizeof(o)), "cfnmpc_default_opts_v")
    for k, v in kw.items():
        if k in ("W", "WN"):
            arr = getattr(o, k)
            for i, x in enumerate(v):
                arr[i] = float(x)
        else:
            setattr(o, k, v)
    return o


def _is_torch(a):
    return type(a).__module__.startswith("torch")


def _stream_ptr(a):
    import torch
    return C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)


def _torch_device():
    """torch's current HIP device if its runtime is up (a solver lives on the device current at its creation), else None"""
    import sys
    torch = sys.modules.get("torch")
    try:
        if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
            return int(torch.cuda.current_device())
    except Exception:
        pass
    return None


def _launch_stream(stream, device=None):
    """Stream of a launch call: the caller's raw hipStream_t, else torch's CURRENT stream on the solver's device (the one
    the device-tensor setters enqueue on: a `with torch.cuda.stream(s):` block keeps setters and solve in one queue),
    else the default stream."""
    if stream:
        return C.c_void_p(stream)
    if device is not None:
        import sys
        torch = sys.modules.get("torch")
        try:
            if torch is not None and torch.cuda.is_initialized():
                return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        except Exception:
            pass
    return C.c_void_p(0)


def _arg(a, shape, dtype=np.float64, device=None):
    """-> (pointer, on_device, stream, keepalive).  Host arrays travel on the stream the launch calls use (torch's CURRENT
    stream on the solver's device, _launch_stream): inside `with torch.cuda.stream(s):` a numpy getter then waits for the
    solve enqueued on s, and a numpy setter cannot overtake it -- pool streams do not synchronise with stream 0."""
    if _is_torch(a):
        import torch
        want = {np.float64: torch.float64, np.int32: torch.int32}[dtype]
        if not a.is_cuda or a.dtype != want or not a.is_contiguous() or tuple(a.shape) != tuple(shape):
            raise ValueError(f"expected contiguous device tensor {shape} {want}, got {tuple(a.shape)} {a.dtype}")
        return C.c_void_p(a.data_ptr()), 1, _stream_ptr(a), a
    arr = np.ascontiguousarray(a, dtype=dtype)
    if arr.shape != tuple(shape):
        raise ValueError(f"expected shape {shape}, got {arr.shape}")
    return arr.ctypes.data_as(C.c_void_p), 0, _launch_stream(None, device), arr


class BatchSolver:
    def __init__(self, batch: int, opts: Opts | None = None, **kw):
        self._L = _lib.lib()
        self.opts = opts if opts is not None else default_opts(**kw)
        self.B = int(batch)
        self.N = int(self.opts.N)
        h = C.c_void_p()
        _check(self._L.cfnmpc_create(C.byref(h), self.B, C.byref(self.opts)), "cfnmpc_create")
        self._h = h
        self._device = _torch_device()

    def close(self):
        if getattr(self, "_h", None):
            self._L.cfnmpc_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def workspace_bytes(self):
        return int(self._L.cfnmpc_workspace_bytes(self._h))

    # ---- inputs
    def set_x0(self, x0):
        p, dev, st, _k = _arg(x0, (self.B, NX), device=self._device)
        _check(self._L.cfnmpc_set_x0(self._h, p, dev, st), "cfnmpc_set_x0")

    def set_yref(self, yref, yref_e):
        p, dev, st, _k = _arg(yref, (self.B, self.N, NY), device=self._device)
        pe, deve, _st, _k2 = _arg(yref_e, (self.B, NX), device=self._device)
        if dev != deve:
            raise ValueError("yref and yref_e must live on the same side")
        _check(self._L.cfnmpc_set_yref(self._h, p, pe, dev, st), "cfnmpc_set_yref")

    def set_yref_windows(self, traj, mode, it, des_xyz, uss):
        """Device-side reference windows (NMPC::iteration state machine); all arguments are torch
        device tensors: traj [n_rows][17] float64 or None, mode / it [B] int32 (updated in place),
        des_xyz [B][3] float64."""
        import torch
        assert mode.is_cuda and it.is_cuda and des_xyz.is_cuda and mode.dtype == torch.int32 and it.dtype == torch.int32
        assert tuple(des_xyz.shape) == (self.B, 3) and des_xyz.is_contiguous() and des_xyz.dtype == torch.float64
        n_rows, tp = 0, None
        if traj is not None:
            assert traj.is_cuda and traj.dtype == torch.float64 and traj.is_contiguous() and traj.shape[1] == 17
            n_rows, tp = int(traj.shape[0]), C.c_void_p(traj.data_ptr())
        _check(self._L.cfnmpc_set_yref_windows(self._h, tp, n_rows, C.c_void_p(mode.data_ptr()), C.c_void_p(it.data_ptr()),
                                               C.c_void_p(des_xyz.data_ptr()), float(uss), _stream_ptr(mode)),
               "cfnmpc_set_yref_windows")

    def set_weights(self, W=None, WN=None):
        w = None if W is None else np.ascontiguousarray(W, dtype=np.float64)
        wn = None if WN is None else np.ascontiguousarray(WN, dtype=np.float64)
        _check(self._L.cfnmpc_set_weights(self._h, None if w is None else w.ctypes.data_as(C.c_void_p),
                                          None if wn is None else wn.ctypes.data_as(C.c_void_p)), "cfnmpc_set_weights")

    def set_box(self, u_min, u_max):
        _check(self._L.cfnmpc_set_box(self._h, float(u_min), float(u_max)), "cfnmpc_set_box")

    def set_box_stages(self, lb=None, ub=None):
        """Per-stage, per-input box [B][N][4] (acados' "lbu" / "ubu" on individual stages); None, None: back to
        the scalar box."""
        if lb is None and ub is None:
            _check(self._L.cfnmpc_set_box_stages(self._h, None, None, 0, _launch_stream(None, self._device)), "cfnmpc_set_box_stages")
            return
        p, dev, st, _k = _arg(lb, (self.B, self.N, NU), device=self._device)
        pu, devu, _s, _k2 = _arg(ub, (self.B, self.N, NU), device=self._device)
        assert dev == devu
        _check(self._L.cfnmpc_set_box_stages(self._h, p, pu, dev, st), "cfnmpc_set_box_stages")

    def init_iterate(self, mode=INIT_ACADOS, stream=None):
        _check(self._L.cfnmpc_init_iterate(self._h, mode, _launch_stream(stream, self._device)), "cfnmpc_init_iterate")

    def set_iterate(self, x, u):
        p, dev, st, _k = _arg(x, (self.B, self.N + 1, NX), device=self._device)
        pu, devu, _s, _k2 = _arg(u, (self.B, self.N, NU), device=self._device)
        assert dev == devu
        _check(self._L.cfnmpc_set_iterate(self._h, p, pu, dev, st), "cfnmpc_set_iterate")

    # ---- solve
    def solve(self, n_rti=1, stream=None):
        """acados_solve() for the batch; `stream` is a raw hipStream_t (int) or None = default."""
        _check(self._L.cfnmpc_solve(self._h, int(n_rti), _launch_stream(stream, self._device)), "cfnmpc_solve")

    def step_host(self, x0, yref, yref_e, stream=None):
        """cfnmpc_step_host: host arrays in (x0 [B,13], yref [B,N,17], yref_e [B,13]), one RTI step,
        host arrays out -> (u [B,N,4], x [B,N+1,13], status, qp_iter, res); one synchronisation."""
        x0 = np.ascontiguousarray(x0, dtype=np.float64); yref = np.ascontiguousarray(yref, dtype=np.float64)
        yref_e = np.ascontiguousarray(yref_e, dtype=np.float64)
        assert x0.shape == (self.B, NX) and yref.shape == (self.B, self.N, NY) and yref_e.shape == (self.B, NX)
        u = np.empty((self.B, self.N, NU)); x = np.empty((self.B, self.N + 1, NX))
        st = np.empty(self.B, dtype=np.int32); it = np.empty(self.B, dtype=np.int32); rs = np.empty(self.B)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        _check(self._L.cfnmpc_step_host(self._h, vp(x0), vp(yref), vp(yref_e), vp(u), vp(x), vp(st), vp(it), vp(rs),
                                        _launch_stream(stream, self._device)), "cfnmpc_step_host")
        return u, x, st, it, rs

    def set_profiling(self, enable=True):
        _check(self._L.cfnmpc_set_profiling(self._h, int(bool(enable))), "cfnmpc_set_profiling")

    def get_profile(self):
        """-> (ms_linearise, ms_qp, n_steps): average kernel durations since the last call."""
        a = C.c_double(0); b = C.c_double(0); n = C.c_int(0)
        _check(self._L.cfnmpc_get_profile(self._h, C.byref(a), C.byref(b), C.byref(n)), "cfnmpc_get_profile")
        return a.value, b.value, n.value

    def get_profile_kernels(self):
        """-> (ms[6], n_steps): linearise | factor | forward | compaction | active set | interior point (averages)."""
        ms = (C.c_double * 6)(); n = C.c_int(0)
        _check(self._L.cfnmpc_get_profile_kernels(self._h, ms, C.byref(n)), "cfnmpc_get_profile_kernels")
        return [float(v) for v in ms], n.value

    def get_profile_steps(self, max_steps=4096):
        """-> array [n_steps][6]: the six kernel-group durations of every timed step (ms), not averaged."""
        ms = np.zeros((int(max_steps), 6)); n = C.c_int(0)
        _check(self._L.cfnmpc_get_profile_steps(self._h, ms.ctypes.data_as(C.c_void_p), int(max_steps), C.byref(n)), "cfnmpc_get_profile_steps")
        return ms[:n.value].copy()

    def linearise_only(self, stream=None):
        _check(self._L.cfnmpc_debug_linearise(self._h, _launch_stream(stream, self._device)), "cfnmpc_debug_linearise")

    def start_factor(self, mode, reps=1, stream=None):
        """Backward half of the start solve only (development / parity tests): mode 1 = k_linearise + k_factor,
        2 = the fused k_linfactor.  Returns the average duration of one repetition [ms]."""
        ms = C.c_double(0.0)
        _check(self._L.cfnmpc_debug_start_factor(self._h, int(mode), int(reps), C.byref(ms), _launch_stream(stream, self._device)),
               "cfnmpc_debug_start_factor")
        return ms.value

    def get_factor(self):
        """-> K [B][N][4][13], d [B][N][4], Pchk [B][6][13][13], status [B] of the last start solve (reference state order)"""
        K = np.empty((self.B, self.N, 4, NX)); d = np.empty((self.B, self.N, 4)); Pc = np.zeros((self.B, 6, NX, NX))
        st = np.empty(self.B, dtype=np.int32)
        _check(self._L.cfnmpc_debug_get_factor(self._h, K.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p),
                                               Pc.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p)), "cfnmpc_debug_get_factor")
        return K, d, Pc, st

    # ---- outputs
    def get_iterate(self):
        x = np.empty((self.B, self.N + 1, NX)); u = np.empty((self.B, self.N, NU))
        _check(self._L.cfnmpc_get_iterate(self._h, x.ctypes.data_as(C.c_void_p), u.ctypes.data_as(C.c_void_p), 0,
                                          _launch_stream(None, self._device)), "cfnmpc_get_iterate")
        return x, u

    def get_u(self, stage, out=None):
        if out is None:
            out = np.empty((self.B, NU))
        p, dev, st, _k = _arg(out, (self.B, NU), device=self._device)
        _check(self._L.cfnmpc_get_u(self._h, int(stage), p, dev, st), "cfnmpc_get_u")
        return out

    def get_x(self, stage, out=None):
        if out is None:
            out = np.empty((self.B, NX))
        p, dev, st, _k = _arg(out, (self.B, NX), device=self._device)
        _check(self._L.cfnmpc_get_x(self._h, int(stage), p, dev, st), "cfnmpc_get_x")
        return out

    def get_cmd(self, cmd_vel=None, motvel=None):
        """Output stage of the reference node for the fleet on the device (cfnmpc_get_cmd):
        -> (cmd_vel [B][4] float64 = pitch deg, -roll deg, thrust PWM, yaw rate deg/s; motvel [B][4] int32).
        numpy arrays (host) or torch device tensors."""
        if cmd_vel is None:
            cmd_vel = np.empty((self.B, 4))
        if motvel is None:
            if _is_torch(cmd_vel):
                import torch
                motvel = torch.empty((self.B, 4), dtype=torch.int32, device=cmd_vel.device)
            else:
                motvel = np.empty((self.B, 4), dtype=np.int32)
        p, dev, st, _k = _arg(cmd_vel, (self.B, 4), device=self._device)
        pm, devm, _s, _k2 = _arg(motvel, (self.B, 4), dtype=np.int32, device=self._device)
        assert dev == devm
        _check(self._L.cfnmpc_get_cmd(self._h, p, pm, dev, st), "cfnmpc_get_cmd")
        return cmd_vel, motvel

    def stats(self):
        st = np.empty(self.B, dtype=np.int32); it = np.empty(self.B, dtype=np.int32); rs = np.empty(self.B)
        _check(self._L.cfnmpc_get_stats(self._h, st.ctypes.data_as(C.c_void_p), it.ctypes.data_as(C.c_void_p),
                                        rs.ctypes.data_as(C.c_void_p), 0, _launch_stream(None, self._device)), "cfnmpc_get_stats")
        return st, it, rs

    def get_linearisation(self):
        A = np.empty((self.B, self.N, NX, NX)); Bm = np.empty((self.B, self.N, NX, NU)); b = np.empty((self.B, self.N, NX))
        _check(self._L.cfnmpc_debug_get_linearisation(self._h, A.ctypes.data_as(C.c_void_p), Bm.ctypes.data_as(C.c_void_p),
                                                      b.ctypes.data_as(C.c_void_p)), "cfnmpc_debug_get_linearisation")
        return A, Bm, b

    def get_condensed(self, block):
        """Partial condensing (cond_N2 > 0): condensed block `block` of every instance after a fresh
        linearisation + pcond -> (H [B][w][w], D [B][13][w], m)."""
        N2 = int(self.opts.cond_N2)
        assert 0 < N2 < self.N
        mmax = -(-self.N // N2)
        w = 4 * mmax + 14
        H = np.zeros((self.B, w, w)); D = np.zeros((self.B, NX, w)); m = C.c_int(0)
        _check(self._L.cfnmpc_debug_get_condensed(self._h, int(block), H.ctypes.data_as(C.c_void_p), D.ctypes.data_as(C.c_void_p),
                                                  C.byref(m)), "cfnmpc_debug_get_condensed")
        wj = 4 * m.value + 14
        return (H.reshape(-1)[:self.B * wj * wj].reshape(self.B, wj, wj).copy(),      # (the library packs with the block's own w)
                D.reshape(-1)[:self.B * NX * wj].reshape(self.B, NX, wj).copy(), m.value)

    def list_counts(self):
        """-> (constrained rows listed, rows listed for the interior-point fall-back, listed rows with heads > 16 stages,
        late rows of a split forward sweep) of the last step"""
        c = np.zeros(4, dtype=np.int32)
        _check(self._L.cfnmpc_debug_get_list_counts(self._h, c.ctypes.data_as(C.c_void_p)), "cfnmpc_debug_get_list_counts")
        return tuple(int(v) for v in c)

    def heads(self):
        h = np.empty(self.B, dtype=np.int32)
        _check(self._L.cfnmpc_debug_get_head(self._h, h.ctypes.data_as(C.c_void_p)), "cfnmpc_debug_get_head")
        return h


def sim(x, u, T=0.06, steps=4, out=None):
    """Batched predictor / plant step (crazyflie_acados_sim_solve, acados_estimator.cpp:589)."""
    L = _lib.lib()
    B = x.shape[0]
    if _is_torch(x):
        import torch
        if out is None:
            out = torch.empty_like(x)
        px, _d, st, _k = _arg(x, (B, NX), device=_torch_device()); pu, _d2, _s, _k2 = _arg(u, (B, NU), device=_torch_device()); po, _d3, _s3, _k3 = _arg(out, (B, NX), device=_torch_device())
        _check(L.cfnmpc_sim(B, px, pu, float(T), int(steps), po, 1, st), "cfnmpc_sim")
        return out
    xa = np.ascontiguousarray(x, dtype=np.float64); ua = np.ascontiguousarray(u, dtype=np.float64)
    if out is None:
        out = np.empty_like(xa)
    _check(L.cfnmpc_sim(B, xa.ctypes.data_as(C.c_void_p), ua.ctypes.data_as(C.c_void_p), float(T), int(steps),
                        out.ctypes.data_as(C.c_void_p), 0, None), "cfnmpc_sim")
    return out


def estimate(meas, filt, u, dt=0.015, use_lpf=True, delay=0.06, steps=4):
    """Batched ESTIMATOR::predictor (acados_estimator.cpp:521-634) on torch device tensors:
    meas [B][9], filt [B][9] (updated in place), u [B][4] -> (x_est, x_pred) [B][13]."""
    import torch
    L = _lib.lib()
    B = meas.shape[0]
    for a, sh in ((meas, (B, 9)), (filt, (B, 9)), (u, (B, 4))):
        assert a.is_cuda and a.dtype == torch.float64 and a.is_contiguous() and tuple(a.shape) == sh
    x_est = torch.empty((B, NX), dtype=torch.float64, device=meas.device)
    x_pred = torch.empty_like(x_est)
    _check(L.cfnmpc_estimate(B, C.c_void_p(meas.data_ptr()), C.c_void_p(filt.data_ptr()), C.c_void_p(u.data_ptr()),
                             float(dt), int(bool(use_lpf)), float(delay), int(steps), C.c_void_p(x_est.data_ptr()),
                             C.c_void_p(x_pred.data_ptr()), _stream_ptr(meas)), "cfnmpc_estimate")
    return x_est, x_pred
