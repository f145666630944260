"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/*.h declares.  No compute calls (no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", src)
    return sorted(set(n for n in names if not n.isupper()))


def test_batch_abi_symbols_exported():
    from crazyflie_nmpc_amd import _lib
    L = _lib.lib()
    declared = _declared("cfnmpc.h")
    assert set(declared) == set(_lib.SYMBOLS), (declared, _lib.SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name


def test_default_opts_match_reference_generator():
    # generate_c_code.py:41-42,63-84,109,133-134
    from crazyflie_nmpc_amd import default_opts
    o = default_opts()
    assert o.N == 50 and abs(o.dt - 0.015) < 1e-15
    assert list(o.W) == [120.0, 100.0, 100.0, 1e-3, 1e-3, 1e-3, 1e-3, 0.7, 1.0, 4.0, 1e-5, 1e-5, 10.0, 0.06, 0.06, 0.06, 0.06]
    assert all(abs(a - 50 * b) < 1e-12 for a, b in zip(o.WN, list(o.W)[:13]))
    assert (o.u_min, o.u_max) == (0.0, 22.0)


def test_no_gpu_means_loud_failure():
    """The product has no CPU fallback: creating a solver without a HIP device must fail."""
    import pytest
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("GPU present")
    from crazyflie_nmpc_amd import BatchSolver
    from crazyflie_nmpc_amd.solver import CfnmpcError
    with pytest.raises(CfnmpcError):
        BatchSolver(4)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "crazyflie_nmpc_amd")
    for dirpath, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "cfnmpc_oracle" not in txt and "cfnmpc_ref" not in txt and "cref" not in txt.replace("cref_", ""), f


def test_headers_are_plain_c(tmp_path):
    """include/*.h are the C-ABI: they must compile as C (gcc -std=c99 -pedantic), not only as C++."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include "cfnmpc.h"\n#include "acados_solver_crazyflie.h"\n#include "acados_sim_solver_crazyflie.h"\n'
                   "int use(void) { cfnmpc_opts o; cfnmpc_default_opts(&o); return o.N + (int)sizeof(sim_in) + (int)sizeof(ocp_nlp_out); }\n")
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_exported_debug_entry_points_are_the_documented_ones():
    """The shipped library exports no experiment: every cfnmpc_debug_* symbol of libcfnmpc.so is declared in
    include/cfnmpc.h and named in INTEGRATION.md (kernel-level access for the parity tests); development entry points
    live in `make DEV=1` builds only (csrc/cfnmpc_dev.h)."""
    import re
    import subprocess
    from crazyflie_nmpc_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\b(cfnmpc_debug_\w+)\b", out))
    header = open(os.path.join(ROOT, "include", "cfnmpc.h")).read()
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    dev = open(os.path.join(ROOT, "crazyflie_nmpc_amd", "csrc", "cfnmpc_dev.h")).read()
    assert exported, "no debug accessors found (nm output changed?)"
    for name in exported:
        assert name in header, name
        assert name in integ or name.replace("cfnmpc_debug_get_", "cfnmpc_debug_get_") in integ, name
        assert name not in dev, name
    for name in re.findall(r"\b(cfnmpc_debug_\w+)\s*\(", dev):
        assert name not in exported, name
    # and the library reads no environment variable
    assert "getenv" not in subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
