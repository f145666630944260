"""Register / scratch / LDS usage of every kernel of the product library, one line each.

Reads the kernel-resource-usage remarks the build leaves beside each device unit's object
(crazyflie_nmpc_amd/csrc/build/<unit>.res, written by the Makefile's compile rule), so the figures are those of the code that
was actually built -- all four device units (cfnmpc_kernels, cfnmpc_linfactor, cfnmpc_asdense, cfnmpc_pcond).
    python tools/resource.py [kernel-name-substring ...]
tests/test_resource_budget.py asserts the budgets of the hot kernels on the same data (resource_table())."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "crazyflie_nmpc_amd", "csrc")
UNITS = ("cfnmpc_kernels", "cfnmpc_linfactor", "cfnmpc_asdense", "cfnmpc_pcond")
KEYS = {"VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy",
        "LDS Size [bytes/block]": "lds", "VGPRs Spill": "vgpr_spill", "SGPRs Spill": "sgpr_spill", "TotalSGPRs": "sgpr",
        "Dynamic Stack": "dynamic_stack"}


def _demangle(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    clean = []
    for o in out[:len(names)]:
        o = o.replace("(anonymous namespace)::", "").replace("cfn::", "")
        o = o[5:] if o.startswith("void ") else o          # (template instances demangle with their return type)
        clean.append(o.split("(")[0].replace(" ", "").strip())
    return clean


def resource_table(build_dir=None):
    """-> {kernel name: dict(unit, vgpr, agpr, scratch, occupancy, lds, vgpr_spill, sgpr_spill, sgpr)} for every __global__
    function of the four device units; raises FileNotFoundError if a unit's listing is missing (library not built)."""
    build_dir = build_dir or os.path.join(CSRC, "build")
    table = {}
    for unit in UNITS:
        path = os.path.join(build_dir, unit + ".res")
        with open(path) as f:
            text = f.read()
        cur = None
        entries = []
        for ln in text.splitlines():
            m = re.search(r"remark:\s+(.*?) \[-Rpass-analysis=kernel-resource-usage\]", ln)
            if not m:
                continue
            t = m.group(1).strip()
            if t.startswith("Function Name:"):
                cur = {"unit": unit, "_mangled": t.split(":", 1)[1].strip()}
                entries.append(cur)
            elif cur is not None and ":" in t:
                k, v = t.rsplit(":", 1)
                k = k.strip()
                if k in KEYS:
                    v = v.strip()
                    cur[KEYS[k]] = (v == "True") if k == "Dynamic Stack" else int(v)
        for e, name in zip(entries, _demangle([e["_mangled"] for e in entries])):
            del e["_mangled"]
            table[name] = e
    return table


if __name__ == "__main__":
    tab = resource_table()
    pats = sys.argv[1:]
    print(f"{'kernel':28s} {'unit':18s} {'V':>4s} {'A':>4s} {'scratch':>8s} {'spill':>6s} {'occ':>4s} {'LDS':>7s}")
    for name, e in sorted(tab.items(), key=lambda kv: (kv[1]["unit"], kv[0])):
        if pats and not any(p in name for p in pats):
            continue
        print(f"{name:28s} {e['unit']:18s} {e['vgpr']:4d} {e['agpr']:4d} {e['scratch']:8d} {e['vgpr_spill']:6d} {e['occupancy']:4d} {e['lds']:7d}")
