"""ctypes binding of libcfnmpc.so (include/cfnmpc.h).  Fails loudly when the library is absent:
the product has no CPU path."""
from __future__ import annotations

import ctypes as C
import os

NX, NU, NY, NYN = 13, 4, 17, 13
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CFNMPC_LIB") or os.path.join(_HERE, "libcfnmpc.so")   # (CFNMPC_LIB: development aid, A/B builds)

# every symbol include/cfnmpc.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "cfnmpc_default_opts", "cfnmpc_default_opts_v", "cfnmpc_opts_size", "cfnmpc_abi_version", "cfnmpc_create", "cfnmpc_free", "cfnmpc_batch", "cfnmpc_horizon",
    "cfnmpc_workspace_bytes", "cfnmpc_set_x0", "cfnmpc_set_yref", "cfnmpc_set_weights", "cfnmpc_set_box", "cfnmpc_get_cmd", "cfnmpc_set_yref_windows", "cfnmpc_init_iterate",
    "cfnmpc_set_iterate", "cfnmpc_get_iterate", "cfnmpc_solve", "cfnmpc_step_host", "cfnmpc_get_u", "cfnmpc_get_x",
    "cfnmpc_get_stats", "cfnmpc_sim", "cfnmpc_estimate", "cfnmpc_debug_get_linearisation", "cfnmpc_debug_linearise", "cfnmpc_debug_start_factor", "cfnmpc_debug_get_factor",
    "cfnmpc_debug_get_head", "cfnmpc_debug_get_viol", "cfnmpc_debug_get_list_counts", "cfnmpc_debug_get_condensed", "cfnmpc_set_box_stages", "cfnmpc_set_profiling", "cfnmpc_get_profile", "cfnmpc_get_profile_kernels", "cfnmpc_get_profile_steps", "cfnmpc_version",
    "cfnmpc_fleet_create", "cfnmpc_fleet_free", "cfnmpc_fleet_batch", "cfnmpc_fleet_min_horizon", "cfnmpc_fleet_max_horizon",
    "cfnmpc_fleet_num_buckets", "cfnmpc_fleet_bucket", "cfnmpc_fleet_workspace_bytes", "cfnmpc_fleet_set_x0",
    "cfnmpc_fleet_set_yref", "cfnmpc_fleet_set_weights", "cfnmpc_fleet_init_iterate", "cfnmpc_fleet_solve",
    "cfnmpc_fleet_get_u", "cfnmpc_fleet_get_x", "cfnmpc_fleet_get_stats", "cfnmpc_fleet_set_box", "cfnmpc_fleet_set_box_stages", "cfnmpc_fleet_get_cmd",
    "cfnmpc_multi_create", "cfnmpc_multi_free", "cfnmpc_multi_batch", "cfnmpc_multi_num_shards", "cfnmpc_multi_shard",
    "cfnmpc_multi_set_x0", "cfnmpc_multi_set_yref", "cfnmpc_multi_set_weights", "cfnmpc_multi_init_iterate", "cfnmpc_multi_solve",
    "cfnmpc_multi_sync", "cfnmpc_multi_set_box", "cfnmpc_multi_set_box_stages", "cfnmpc_multi_get_u", "cfnmpc_multi_get_x", "cfnmpc_multi_get_cmd", "cfnmpc_multi_get_stats",
    "cfnmpc_multi_create_horizons", "cfnmpc_multi_shard_fleet", "cfnmpc_shard_by_horizon",
]
ABI_VERSION = 9   # CFNMPC_ABI_VERSION of the include/cfnmpc.h this binding was written against


class Opts(C.Structure):
    """struct cfnmpc_opts"""
    _fields_ = [("struct_size", C.c_int), ("N", C.c_int), ("dt", C.c_double), ("W", C.c_double * NY), ("WN", C.c_double * NYN),
                ("u_min", C.c_double), ("u_max", C.c_double), ("tol", C.c_double),
                ("max_iter", C.c_int), ("tau", C.c_double), ("thr0", C.c_double),
                ("lam0_min", C.c_double), ("mu0_scale", C.c_double), ("active_horizon", C.c_int), ("ah_margin", C.c_double),
                ("ah_extra", C.c_int), ("active_set", C.c_int), ("forward_sweep", C.c_int), ("cond_N2", C.c_int), ("step_graph", C.c_int), ("as_passes", C.c_int), ("ipm_clip_viol", C.c_double), ("ipm_clip_margin", C.c_double), ("as_skip_viol", C.c_double), ("reinit_failed", C.c_int), ("start_solve", C.c_int), ("as_warm", C.c_int), ("as_dense", C.c_int), ("forward_split", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C crazyflie_nmpc_amd/csrc). "
            "crazyflie_nmpc_amd has no CPU fallback.")
    # PyTorch wheels bundle their own HIP runtime (same soname as /opt/rocm's).  If this library
    # initialised HIP first, a later `import torch` would bring up a second runtime instance and
    # find no GPU; loading torch first makes both share one instance (torch is plumbing only).
    try:
        import torch  # noqa: F401
    except Exception:  # torch absent: the library runs on the system ROCm runtime alone
        pass
    L = C.CDLL(LIB_PATH)
    vp, ip, dbl, i32 = C.c_void_p, C.c_int, C.c_double, C.c_int
    # ABI guard (include/cfnmpc.h): this binding's struct must be the library's before anything is written through it
    if not hasattr(L, "cfnmpc_opts_size") or L.cfnmpc_opts_size() != C.sizeof(Opts) or L.cfnmpc_abi_version() != ABI_VERSION:
        have = (L.cfnmpc_opts_size(), L.cfnmpc_abi_version()) if hasattr(L, "cfnmpc_opts_size") else "no ABI guard (older library)"
        raise ImportError(f"{LIB_PATH}: cfnmpc_opts size / ABI version {have} do not match this binding's "
                          f"({C.sizeof(Opts)}, {ABI_VERSION}): rebuild the library (make -C crazyflie_nmpc_amd/csrc)")
    L.cfnmpc_default_opts.argtypes = [C.POINTER(Opts)]
    L.cfnmpc_default_opts.restype = None
    L.cfnmpc_default_opts_v.argtypes = [C.POINTER(Opts), i32]
    L.cfnmpc_multi_create_horizons.argtypes = [C.POINTER(vp), i32, vp, i32, vp, C.POINTER(Opts)]
    L.cfnmpc_multi_shard_fleet.argtypes = [vp, i32, C.POINTER(vp), vp, vp, vp, C.POINTER(vp)]
    L.cfnmpc_shard_by_horizon.argtypes = [i32, vp, i32, vp]
    L.cfnmpc_create.argtypes = [C.POINTER(vp), i32, C.POINTER(Opts)]
    L.cfnmpc_free.argtypes = [vp]
    L.cfnmpc_batch.argtypes = [vp]
    L.cfnmpc_horizon.argtypes = [vp]
    L.cfnmpc_workspace_bytes.argtypes = [vp]
    L.cfnmpc_workspace_bytes.restype = C.c_ulonglong
    L.cfnmpc_set_x0.argtypes = [vp, vp, i32, vp]
    L.cfnmpc_set_yref.argtypes = [vp, vp, vp, i32, vp]
    L.cfnmpc_set_weights.argtypes = [vp, vp, vp]
    L.cfnmpc_set_box.argtypes = [vp, dbl, dbl]
    L.cfnmpc_get_cmd.argtypes = [vp, vp, vp, i32, vp]
    L.cfnmpc_set_yref_windows.argtypes = [vp, vp, i32, vp, vp, vp, dbl, vp]
    L.cfnmpc_init_iterate.argtypes = [vp, i32, vp]
    L.cfnmpc_set_iterate.argtypes = [vp, vp, vp, i32, vp]
    L.cfnmpc_get_iterate.argtypes = [vp, vp, vp, i32, vp]
    L.cfnmpc_solve.argtypes = [vp, i32, vp]
    L.cfnmpc_step_host.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.cfnmpc_get_u.argtypes = [vp, i32, vp, i32, vp]
    L.cfnmpc_get_x.argtypes = [vp, i32, vp, i32, vp]
    L.cfnmpc_get_stats.argtypes = [vp, vp, vp, vp, i32, vp]
    L.cfnmpc_sim.argtypes = [i32, vp, vp, dbl, i32, vp, i32, vp]
    L.cfnmpc_estimate.argtypes = [i32, vp, vp, vp, dbl, i32, dbl, i32, vp, vp, vp]
    L.cfnmpc_debug_get_linearisation.argtypes = [vp, vp, vp, vp]
    L.cfnmpc_debug_get_head.argtypes = [vp, vp]
    L.cfnmpc_debug_get_list_counts.argtypes = [vp, vp]
    L.cfnmpc_debug_get_condensed.argtypes = [vp, i32, vp, vp, vp]
    L.cfnmpc_set_profiling.argtypes = [vp, i32]
    L.cfnmpc_get_profile.argtypes = [vp, vp, vp, vp]
    L.cfnmpc_set_box_stages.argtypes = [vp, vp, vp, i32, vp]
    L.cfnmpc_get_profile_kernels.argtypes = [vp, vp, vp]
    L.cfnmpc_get_profile_steps.argtypes = [vp, vp, i32, vp]
    L.cfnmpc_debug_linearise.argtypes = [vp, vp]
    for name, at in (("cfnmpc_debug_chunked_pair", [vp, i32, i32, vp, vp]), ("cfnmpc_debug_checksum", [vp, vp]),
                     ("cfnmpc_debug_solve_part", [vp, i32, i32, vp])):   # development builds only (make DEV=1, csrc/cfnmpc_dev.h)
        if hasattr(L, name):
            getattr(L, name).argtypes = at
    L.cfnmpc_debug_start_factor.argtypes = [vp, i32, i32, vp, vp]
    L.cfnmpc_debug_get_factor.argtypes = [vp, vp, vp, vp, vp]
    L.cfnmpc_version.restype = C.c_char_p
    L.cfnmpc_fleet_create.argtypes = [C.POINTER(vp), i32, vp, C.POINTER(Opts)]
    for n in ("free", "batch", "min_horizon", "max_horizon", "num_buckets", "workspace_bytes"):
        getattr(L, "cfnmpc_fleet_" + n).argtypes = [vp]
    L.cfnmpc_fleet_workspace_bytes.restype = C.c_ulonglong
    L.cfnmpc_fleet_bucket.argtypes = [vp, i32, vp, vp, C.POINTER(vp), vp]
    L.cfnmpc_fleet_set_x0.argtypes = [vp, vp, i32, vp]
    L.cfnmpc_fleet_set_yref.argtypes = [vp, vp, vp, i32, vp]
    L.cfnmpc_fleet_set_weights.argtypes = [vp, vp, vp]
    L.cfnmpc_fleet_set_box.argtypes = [vp, dbl, dbl]
    L.cfnmpc_fleet_set_box_stages.argtypes = [vp, vp, vp]
    L.cfnmpc_fleet_get_cmd.argtypes = [vp, vp, vp, i32, vp]
    L.cfnmpc_fleet_init_iterate.argtypes = [vp, i32, vp]
    L.cfnmpc_fleet_solve.argtypes = [vp, i32, vp]
    L.cfnmpc_fleet_get_u.argtypes = [vp, i32, vp, i32, vp]
    L.cfnmpc_fleet_get_x.argtypes = [vp, i32, vp, i32, vp]
    L.cfnmpc_fleet_get_stats.argtypes = [vp, vp, vp, vp, i32, vp]
    L.cfnmpc_multi_create.argtypes = [C.POINTER(vp), i32, vp, i32, C.POINTER(Opts)]
    for n in ("free", "batch", "num_shards", "sync"):
        getattr(L, "cfnmpc_multi_" + n).argtypes = [vp]
    L.cfnmpc_multi_shard.argtypes = [vp, i32, C.POINTER(vp), vp, vp, vp, C.POINTER(vp)]
    L.cfnmpc_multi_set_x0.argtypes = [vp, vp]
    L.cfnmpc_multi_set_yref.argtypes = [vp, vp, vp]
    L.cfnmpc_multi_set_weights.argtypes = [vp, vp, vp]
    L.cfnmpc_multi_init_iterate.argtypes = [vp, i32]
    L.cfnmpc_multi_solve.argtypes = [vp, i32]
    L.cfnmpc_multi_get_u.argtypes = [vp, i32, vp]
    L.cfnmpc_multi_get_x.argtypes = [vp, i32, vp]
    L.cfnmpc_multi_get_cmd.argtypes = [vp, vp, vp]
    L.cfnmpc_multi_get_stats.argtypes = [vp, vp, vp, vp]
    L.cfnmpc_multi_set_box.argtypes = [vp, dbl, dbl]
    L.cfnmpc_multi_set_box_stages.argtypes = [vp, vp, vp]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if fn.restype is C.c_int or fn.restype is None or name in ("cfnmpc_version", "cfnmpc_workspace_bytes"):
            continue
    _lib = L
    return L
