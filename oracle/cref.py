"""ctypes binding of oracle/libcfnmpc_oracle.so (the plain-C CPU restatement).
TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcfnmpc_oracle.so")

NX, NU, NY = 13, 4, 17


class Opts(C.Structure):
    _fields_ = [("N", C.c_int), ("dt", C.c_double), ("W", C.c_double * NY), ("WN", C.c_double * NX),
                ("u_min", C.c_double), ("u_max", C.c_double), ("tol", C.c_double),
                ("max_iter", C.c_int), ("tau", C.c_double), ("thr0", C.c_double),
                ("lam0_min", C.c_double), ("mu0_scale", C.c_double), ("active_set", C.c_int),
                ("clip_viol", C.c_double), ("clip_margin", C.c_double), ("as_skip_viol", C.c_double), ("as_warm", C.c_int)]


def _host_tag():
    """What `-march=native` resolves to depends on the CPU the library is BUILT on: the tag (CPU model + ISA
    flags) is stored beside the .so so that a library built on another host (the .so travels with the
    repository snapshot) is rebuilt before it is loaded or timed here."""
    model, flags = "", ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and not model:
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("flags") and not flags:
                flags = ln.split(":", 1)[1].strip()
            if model and flags:
                break
    except OSError:
        pass
    import hashlib
    return model + " #" + hashlib.sha1(flags.encode()).hexdigest()[:12]


def march_native():
    """The -march the compiler resolves `native` to on this host (reported in bench.py's cpu_baseline.sample)."""
    try:
        out = subprocess.run(["gcc", "-march=native", "-Q", "--help=target"], capture_output=True, text=True, timeout=20).stdout
        for ln in out.splitlines():
            if ln.strip().startswith("-march="):
                return ln.split("=", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def build(force=False):
    stamp = _SO + ".host"
    tag = _host_tag()
    stale = (not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "cfnmpc_ref.c"))
             or not os.path.exists(stamp) or open(stamp).read() != tag)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "-s"])
        with open(stamp, "w") as f:
            f.write(tag)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()   # (no-op when the library is current AND was built on this host)
        _lib = C.CDLL(_SO)
        _lib.cfo_rti_step.restype = C.c_int
        _lib.cfo_rti_step_w.restype = C.c_int
        _lib.cfo_closed_loop.restype = C.c_int
        _lib.cfo_qp_solve.restype = C.c_int
    return _lib


def default_opts(N=50, **kw):
    o = Opts()
    lib().cfo_default_opts(C.byref(o))
    o.N = N
    for k, v in kw.items():
        if k in ("W", "WN"):          # array members: element-wise
            arr = getattr(o, k)
            for i, x in enumerate(v):
                arr[i] = float(x)
        else:
            setattr(o, k, v)
    return o


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def f(x, u):
    x = np.ascontiguousarray(x, dtype=np.float64); u = np.ascontiguousarray(u, dtype=np.float64)
    out = np.empty(NX)
    lib().cfo_f(_p(x), _p(u), _p(out))
    return out


def jac(x, u):
    x = np.ascontiguousarray(x, dtype=np.float64); u = np.ascontiguousarray(u, dtype=np.float64)
    J = np.empty((NX, NY))
    lib().cfo_jac(_p(x), _p(u), _p(J))
    return J


def rk4_sens(x, u, dt=0.015):
    x = np.ascontiguousarray(x, dtype=np.float64); u = np.ascontiguousarray(u, dtype=np.float64)
    phi, A, B = np.empty(NX), np.empty((NX, NX)), np.empty((NX, NU))
    lib().cfo_rk4_sens(_p(x), _p(u), C.c_double(dt), _p(phi), _p(A), _p(B))
    return phi, A, B


def sim(x, u, T=0.06, steps=4):
    x = np.ascontiguousarray(x, dtype=np.float64); u = np.ascontiguousarray(u, dtype=np.float64)
    xn = np.empty_like(x)
    lib().cfo_sim(C.c_int(x.shape[0]), _p(x), _p(u), C.c_double(T), C.c_int(steps), _p(xn))
    return xn


def warm_state(B, N):
    """(wcls [B][N][4] uint8, wvalid [B] int32), zeroed: the persistent state of cfo_opts.as_warm for rti_step(warm=...)"""
    return np.zeros((B, N, NU), dtype=np.uint8), np.zeros(B, dtype=np.int32)


def rti_step(opts, x_it, u_it, x0, yref, yref_e, nthreads=1, warm=None):
    """In-place RTI step on C-contiguous float64 arrays.  Returns (status, iters, res, threads).
    warm = warm_state(B, N), kept by the caller from step to step (used when opts.as_warm)."""
    B = x0.shape[0]
    for a in (x_it, u_it, x0, yref, yref_e):
        assert a.dtype == np.float64 and a.flags.c_contiguous
    status = np.empty(B, dtype=np.int32); iters = np.empty(B, dtype=np.int32); res = np.empty(B)
    if warm is not None:
        wc, wv = warm
        assert wc.dtype == np.uint8 and wc.shape == (B, opts.N, NU) and wv.dtype == np.int32 and wv.shape == (B,)
        used = lib().cfo_rti_step_w(C.byref(opts), C.c_int(B), _p(x_it), _p(u_it), _p(x0), _p(yref), _p(yref_e),
                                    _p(status), _p(iters), _p(res), C.c_int(nthreads), _p(wc), _p(wv))
        return status, iters, res, used
    used = lib().cfo_rti_step(C.byref(opts), C.c_int(B), _p(x_it), _p(u_it), _p(x0), _p(yref), _p(yref_e),
                              _p(status), _p(iters), _p(res), C.c_int(nthreads))
    return status, iters, res, used


def linearise(opts, x_it, u_it, x0, yref, yref_e):
    N = opts.N
    A = np.empty((N, NX, NX)); Bm = np.empty((N, NX, NU)); b = np.empty((N, NX))
    q = np.empty((N + 1, NX)); r = np.empty((N, NU))
    lib().cfo_linearise(C.byref(opts), _p(x_it), _p(u_it), _p(x0), _p(yref), _p(yref_e), _p(A), _p(Bm), _p(b), _p(q), _p(r))
    return A, Bm, b, q, r


def qp_solve(opts, x_it, u_it, x0, yref, yref_e):
    N = opts.N
    dx = np.empty((N + 1, NX)); du = np.empty((N, NU)); ll = np.empty((N, NU)); lu = np.empty((N, NU))
    it = C.c_int(0); res = C.c_double(0)
    st = lib().cfo_qp_solve(C.byref(opts), _p(x_it), _p(u_it), _p(x0), _p(yref), _p(yref_e), _p(dx), _p(du), _p(ll), _p(lu), C.byref(it), C.byref(res))
    return dict(dx=dx, du=du, lam_l=ll, lam_u=lu, status=st, iters=it.value, res=res.value)


def closed_loop(opts, x, yref, yref_e, steps, nthreads=0, latencies=False):
    """Per-instance closed loops inside one parallel region (bench.py cpu_baseline).  x [B][13] is
    updated in place.  -> dict(seconds, threads, iters, bad, lat_us)"""
    B = x.shape[0]
    for a in (x, yref, yref_e):
        assert a.dtype == np.float64 and a.flags.c_contiguous
    sec = C.c_double(0); its = C.c_longlong(0); bad = C.c_longlong(0)
    lat = np.zeros(steps) if (latencies and B == 1) else None
    used = lib().cfo_closed_loop(C.byref(opts), C.c_int(B), _p(x), _p(yref), _p(yref_e), C.c_int(steps), C.c_int(nthreads),
                                 C.byref(sec), C.byref(its), C.byref(bad), _p(lat) if lat is not None else None)
    return dict(seconds=sec.value, threads=int(used), iters=int(its.value), bad=int(bad.value), lat_us=lat)
