"""Multi-GPU plumbing: NMPC instances are independent, so a batch shards
across ranks with NO data-path collective; torch.distributed (RCCL on GPUs, gloo in the CPU
tests) is used only to aggregate the report."""
from __future__ import annotations

import numpy as np

BASE_SEED = 20200103


def shard_seed(rank: int) -> int:
    """Every rank draws its own shard of the synthetic fleet."""
    return BASE_SEED + int(rank)


def shard_range(total: int, rank: int, world: int):
    """Contiguous index range of `rank` when `total` instances are split over `world` ranks."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_by_horizon(horizons, world: int):
    """Mixed-horizon batches (config C5): bucket by N (wave-homogeneous), then deal the buckets out so that sum(N_i) -- the
    cost model of a step -- is balanced (SURVEY.md section 8e "Partitioning").  The partitioner is the library's
    (cfnmpc_shard_by_horizon, host code: the same one cfnmpc_multi_create_horizons uses for its shards), so bench.py's ranks
    and an in-process MultiGpuFleet split a fleet identically.  Returns a list of ascending index arrays, one per rank."""
    import ctypes as C
    from . import _lib
    hz = np.ascontiguousarray(horizons, dtype=np.int32)
    of = np.empty(len(hz), dtype=np.int32)
    rc = _lib.lib().cfnmpc_shard_by_horizon(len(hz), hz.ctypes.data_as(C.c_void_p), int(world), of.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError(f"cfnmpc_shard_by_horizon failed with code {rc} (horizons must be >= 1, world >= 1)")
    return [np.nonzero(of == r)[0].astype(np.int64) for r in range(int(world))]


def aggregate_report(elapsed: float, sums, dist=None, device=None):
    """max-over-ranks of the timed region and sum-over-ranks of additive statistics.
    `dist` is torch.distributed (initialised) or None for a single process."""
    sums = np.asarray(sums, dtype=np.float64)
    if dist is None or not dist.is_initialized():
        return float(elapsed), sums
    # (a one-rank group still goes through the collectives: that is the smoke test of the backend under torchrun)
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    s = torch.tensor(sums, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return float(t.item()), s.cpu().numpy()


class MultiGpuFleet:
    """One fleet over several GPUs from ONE process (C-ABI cfnmpc_multi_*): contiguous shards, one
    solver + stream per shard, every shard's step launched before anyone waits.  Host arrays cover
    the whole fleet.  (bench.py keeps the one-process-per-GPU form the driver launches; this is the
    deployment form for a single controller process.)"""

    def __init__(self, total_batch, device_ids, opts=None, horizons=None):
        """horizons [total_batch] (optional): one horizon per vehicle -- config C5; the shards are then cfnmpc_fleets over the
        index sets of shard_by_horizon, host arrays use the fleet layouts (yref [B][Nmax][17], boxes [B][Nmax][4])."""
        import ctypes as C
        from . import _lib
        from .solver import _check, default_opts
        self._C, self._check = C, _check
        self._L = _lib.lib()
        self.B = int(total_batch)
        self.opts = opts if opts is not None else default_opts()
        ids = np.ascontiguousarray(device_ids, dtype=np.int32)
        h = C.c_void_p()
        self.mixed = horizons is not None
        if self.mixed:
            self.horizons = np.ascontiguousarray(horizons, dtype=np.int32)
            assert self.horizons.shape == (self.B,)
            self.N = int(self.horizons.max())       # row stride of yref / boxes
            _check(self._L.cfnmpc_multi_create_horizons(C.byref(h), len(ids), ids.ctypes.data_as(C.c_void_p), self.B,
                                                        self.horizons.ctypes.data_as(C.c_void_p), C.byref(self.opts)),
                   "cfnmpc_multi_create_horizons")
        else:
            self.N = int(self.opts.N)
            _check(self._L.cfnmpc_multi_create(C.byref(h), len(ids), ids.ctypes.data_as(C.c_void_p), self.B, C.byref(self.opts)),
                   "cfnmpc_multi_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.cfnmpc_multi_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def shards(self):
        """-> [(lo, hi, device)]; mixed horizons: [(vehicle indices, device)]"""
        C = self._C
        out = []
        if self.mixed:
            for i in range(self._L.cfnmpc_multi_num_shards(self._h)):
                cnt, dev = C.c_int(0), C.c_int(0)
                self._check(self._L.cfnmpc_multi_shard_fleet(self._h, i, None, C.byref(cnt), None, C.byref(dev), None), "cfnmpc_multi_shard_fleet")
                idx = np.empty(cnt.value, dtype=np.int32)
                self._check(self._L.cfnmpc_multi_shard_fleet(self._h, i, None, None, idx.ctypes.data_as(C.c_void_p), None, None), "cfnmpc_multi_shard_fleet")
                out.append((idx, dev.value))
            return out
        for i in range(self._L.cfnmpc_multi_num_shards(self._h)):
            lo, hi, dev = C.c_int(0), C.c_int(0), C.c_int(0)
            self._check(self._L.cfnmpc_multi_shard(self._h, i, None, C.byref(lo), C.byref(hi), C.byref(dev), None), "cfnmpc_multi_shard")
            out.append((lo.value, hi.value, dev.value))
        return out

    def _p(self, a, shape, dtype=np.float64):
        a = np.ascontiguousarray(a, dtype=dtype)
        assert a.shape == tuple(shape), (a.shape, shape)
        return a, a.ctypes.data_as(self._C.c_void_p)

    def set_x0(self, x0):
        a, p = self._p(x0, (self.B, 13))
        self._check(self._L.cfnmpc_multi_set_x0(self._h, p), "cfnmpc_multi_set_x0")

    def set_yref(self, yref, yref_e):
        a, p = self._p(yref, (self.B, self.N, 17)); b, q = self._p(yref_e, (self.B, 13))
        self._check(self._L.cfnmpc_multi_set_yref(self._h, p, q), "cfnmpc_multi_set_yref")

    def set_box(self, u_min, u_max):
        assert self._L.cfnmpc_multi_set_box(self._h, float(u_min), float(u_max)) == 0

    def set_box_stages(self, lb=None, ub=None):
        """per-stage / per-input boxes [B][N][4] of the whole fleet; None, None: back to the scalar box"""
        if lb is None and ub is None:
            assert self._L.cfnmpc_multi_set_box_stages(self._h, None, None) == 0
            return
        a, p = self._p(lb, (self.B, self.N, 4)); b, q = self._p(ub, (self.B, self.N, 4))
        assert self._L.cfnmpc_multi_set_box_stages(self._h, p, q) == 0

    def init_iterate(self, mode):
        self._check(self._L.cfnmpc_multi_init_iterate(self._h, int(mode)), "cfnmpc_multi_init_iterate")

    def solve(self, n_rti=1):
        self._check(self._L.cfnmpc_multi_solve(self._h, int(n_rti)), "cfnmpc_multi_solve")

    def sync(self):
        self._check(self._L.cfnmpc_multi_sync(self._h), "cfnmpc_multi_sync")

    def get_u(self, stage):
        u = np.empty((self.B, 4))
        self._check(self._L.cfnmpc_multi_get_u(self._h, int(stage), u.ctypes.data_as(self._C.c_void_p)), "cfnmpc_multi_get_u")
        return u

    def get_x(self, stage):
        x = np.empty((self.B, 13))
        self._check(self._L.cfnmpc_multi_get_x(self._h, int(stage), x.ctypes.data_as(self._C.c_void_p)), "cfnmpc_multi_get_x")
        return x

    def get_cmd(self):
        c = np.empty((self.B, 4)); mv = np.empty((self.B, 4), dtype=np.int32)
        self._check(self._L.cfnmpc_multi_get_cmd(self._h, c.ctypes.data_as(self._C.c_void_p), mv.ctypes.data_as(self._C.c_void_p)), "cfnmpc_multi_get_cmd")
        return c, mv

    def stats(self):
        st = np.empty(self.B, dtype=np.int32); it = np.empty(self.B, dtype=np.int32); rs = np.empty(self.B)
        vp = self._C.c_void_p
        self._check(self._L.cfnmpc_multi_get_stats(self._h, st.ctypes.data_as(vp), it.ctypes.data_as(vp), rs.ctypes.data_as(vp)), "cfnmpc_multi_get_stats")
        return st, it, rs
