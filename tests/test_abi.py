"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/*.h declares.  No compute calls (no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"^[ \t]*#.*$", "", src, flags=re.M)      # preprocessor lines (function-like macros are not exported symbols)
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
        assert name in integ, name
        assert name not in dev, name
    for name in re.findall(r"\b(cfnmpc_debug_\w+)\s*\(", dev):
        assert name not in exported, name
    # and the library reads no environment variable
    assert "getenv" not in subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout


def _header_opts_fields():
    """[(name, ctype, array length or 0)] parsed out of `typedef struct cfnmpc_opts { ... }` in include/cfnmpc.h"""
    src = open(os.path.join(ROOT, "include", "cfnmpc.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    body = re.search(r"typedef struct cfnmpc_opts \{(.*?)\} cfnmpc_opts;", src, flags=re.S).group(1)
    consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define (CFNMPC_\w+) (\d+)", src)}
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ctype, rest = decl.split(None, 1)
        for item in rest.split(","):
            m = re.fullmatch(r"\s*(\w+)\s*(?:\[(\w+)\])?\s*", item)
            out.append((m.group(1), ctype, consts.get(m.group(2), 0) if m.group(2) else 0))
    return out


def test_opts_struct_header_binding_and_documented_stub_agree():
    """cfnmpc_opts grows with the engine: the header, the ctypes binding (_lib.Opts) and the stub printed in INTEGRATION.md
    section 3.3 must list the same fields in the same order with the same types, and the library must report that size."""
    from crazyflie_nmpc_amd import _lib
    hdr = _header_opts_fields()
    kinds = {"int": ctypes.c_int, "double": ctypes.c_double}
    want = [(n, kinds[t] * ln if ln else kinds[t]) for n, t, ln in hdr]
    have = list(_lib.Opts._fields_)
    assert [n for n, _ in want] == [n for n, _ in have]
    for (n, tw), (_n, th) in zip(want, have):
        assert ctypes.sizeof(tw) == ctypes.sizeof(th) and (tw is th or getattr(tw, "_length_", 0) == getattr(th, "_length_", 0)), n
    assert hdr[0][0] == "struct_size"
    L = _lib.lib()
    assert L.cfnmpc_opts_size() == ctypes.sizeof(_lib.Opts)
    abi = int(re.search(r"#define CFNMPC_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "cfnmpc.h")).read()).group(1))
    assert L.cfnmpc_abi_version() == abi == _lib.ABI_VERSION
    # the documented stub: same names, same order, same types
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blk = doc[doc.index("class Opts(C.Structure)"):]
    blk = blk[:blk.index("]\n") + 1]
    doc_fields = re.findall(r'\("(\w+)", C\.c_(int|double)(?:\*(\d+))?\)', blk)
    assert [(n, t, int(ln or 0)) for n, t, ln in doc_fields] == hdr
    assert f"cfnmpc_abi_version() == {abi}" in doc


def test_abi_guard_refuses_foreign_opts_structs():
    """A caller whose cfnmpc_opts is not the library's gets CFNMPC_EINVAL and nothing is written into its object."""
    from crazyflie_nmpc_amd import _lib
    L = _lib.lib()
    buf = (ctypes.c_ubyte * 1024)(*([0xAB] * 1024))
    short = ctypes.sizeof(_lib.Opts) - 40          # the size round 4's documented stub had
    assert L.cfnmpc_default_opts_v(ctypes.cast(buf, ctypes.POINTER(_lib.Opts)), short) == -1
    assert bytes(buf) == b"\xab" * 1024
    o = _lib.Opts()
    assert L.cfnmpc_default_opts_v(ctypes.byref(o), ctypes.sizeof(o)) == 0 and o.struct_size == ctypes.sizeof(o) and o.N == 50
    # create refuses a struct whose leading size is foreign BEFORE it looks for a device (EINVAL, not EHIP)
    o.struct_size = short
    h = ctypes.c_void_p()
    assert L.cfnmpc_create(ctypes.byref(h), 4, ctypes.byref(o)) == -1 and not h.value
    hz = (ctypes.c_int * 4)(30, 50, 50, 100)
    assert L.cfnmpc_fleet_create(ctypes.byref(h), 4, hz, ctypes.byref(o)) == -1
    ids = (ctypes.c_int * 2)(0, 0)
    assert L.cfnmpc_multi_create(ctypes.byref(h), 2, ids, 4, ctypes.byref(o)) in (-1, -2)
