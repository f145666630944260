"""Register / scratch / LDS usage per kernel (hipcc -Rpass-analysis=kernel-resource-usage), one line each.
    python tools/resource.py [extra hipcc flags...]"""
import re, subprocess, sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for src in ("cfnmpc_kernels.hip", "cfnmpc_pcond.hip"):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=fast",
           "-Rpass-analysis=kernel-resource-usage", "-c", "-x", "hip", os.path.join(R, "crazyflie_nmpc_amd/csrc", src), "-o", "/dev/null"] + sys.argv[1:]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = {}
    for ln in out.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass", ln)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            if cur:
                print(cur)
            name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
            cur = {"kernel": name.split("(")[0].replace("cfn::", "").replace("(anonymous namespace)::", "")}
        elif ":" in t:
            k, v = t.split(":", 1)
            k = k.strip()
            if k in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "VGPR Spill", "SGPRs"):
                cur[k.split(" ")[0]] = v.strip()
    if cur:
        print(cur)
