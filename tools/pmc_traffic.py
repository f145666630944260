"""Derives per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; collected
in SEPARATE runs, kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).

    python tools/pmc_traffic.py <dir with FETCH pass> <dir with WRITE pass> <out.json> <out_summary.csv> [batch]

Units / corrections: both counters are in KB; on gfx950 FETCH_SIZE counts half of the bytes of
coalesced 8-byte-per-lane streaming reads (cross-checked against the byte model of DESIGN.md
section 6), so reads = 2 x FETCH_SIZE x 1024; WRITE_SIZE matches the byte model uncorrected.
Averages are taken over the last third of each kernel's dispatches (steady state of the closed loop)."""
import csv, glob, json, sys, collections


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            if "(anonymous namespace)::" in name:      # templated kernels of cfnmpc_pcond.hip
                name = "cfn::" + name.split("(anonymous namespace)::")[1]
            acc[name.split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v[-max(1, len(v) // 3):]) / max(1, len(v) // 3) for k, v in acc.items()}


def main():
    fdir, wdir, out_json, out_csv = sys.argv[1:5]
    batch = int(sys.argv[5]) if len(sys.argv) > 5 else 65536
    rd, wr = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(rd) | set(wr)):
        if not k.startswith("cfn::"):
            continue
        r, w = 2.0 * rd.get(k, 0.0) * 1024.0, wr.get(k, 0.0) * 1024.0
        kernels[k] = {"read_bytes": r, "write_bytes": w, "total_bytes": r + w}
    qp = sum(v["total_bytes"] for k, v in kernels.items() if k.split("<")[0] in (
        "cfn::k_factor", "cfn::k_forward", "cfn::k_forward_rg", "cfn::k_rank", "cfn::k_ipm_list",
        "cfn::k_as_solves", "cfn::k_ascommit", "cfn::k_ascommit1", "cfn::k_as_retry", "cfn::k_compact", "cfn::k_scatter", "cfn::k_as", "cfn::k_ipm_rest", "cfn::k_ipm",
        "cfn::k_pcond", "cfn::k_cfactor", "cfn::k_cforward", "cfn::k_cipm",
        "cfn::k_linfactor", "cfn::k_linearise_clist", "cfn::k_as_cst", "cfn::k_ipm_rest_cst", "cfn::k_ipm_cst", "cfn::k_forward_half"))
    json.dump({"batch": batch,
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (KB); reads = 2 x FETCH_SIZE "
                       "(gfx950 correction, see tools/pmc_traffic.py); average of the last third of the dispatches of "
                       "`python bench.py --steps 6 --warmup 20 --no-cpu-baseline`",
               "kernels": kernels, "hbm_bytes_per_launch_k_qp": qp,
               "hbm_bytes_per_step": qp + kernels.get("cfn::k_linearise", {}).get("total_bytes", 0.0)},
              open(out_json, "w"), indent=1)
    with open(out_csv, "w") as f:
        f.write("kernel,read_GB,write_GB,total_GB\n")
        for k, v in kernels.items():
            f.write(f"\"{k}\",{v['read_bytes'] / 1e9:.4f},{v['write_bytes'] / 1e9:.4f},{v['total_bytes'] / 1e9:.4f}\n")
        f.write(f"QP phase (factor+forward+compact+scatter+as+ipm_rest | ipm),,,{qp / 1e9:.4f}\n")
        f.write(f"RTI step (linearise + QP phase),,,{(qp + kernels.get('cfn::k_linearise', {}).get('total_bytes', 0.0)) / 1e9:.4f}\n")
    print(open(out_csv).read())


if __name__ == "__main__":
    main()
