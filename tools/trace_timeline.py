"""Development aid: print a per-kernel timeline (start / end in ms, queue) from a rocprofv3
kernel-trace CSV, for a window of the run -- to see which kernels overlap across streams."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
t0w, t1w = float(sys.argv[2]), float(sys.argv[3])  # window in ms relative to the first cfn kernel
ks = [r for r in rows if r["Kernel_Name"].startswith("cfn::k_")]
base = min(int(r["Start_Timestamp"]) for r in ks)
last = max(int(r["End_Timestamp"]) for r in ks)
print("total span ms", (last - base) / 1e6)
for r in sorted(ks, key=lambda r: int(r["Start_Timestamp"])):
    s = (int(r["Start_Timestamp"]) - base) / 1e6; e = (int(r["End_Timestamp"]) - base) / 1e6
    s -= (last - base) / 1e6 - t1w  # window measured back from the end
    e -= (last - base) / 1e6 - t1w
    if e < t0w or s > t1w: continue
    name = r["Kernel_Name"].split("(")[0].replace("cfn::", "")
    if name in ("k_put", "k_get", "k_sim"): continue
    print(f"{s:9.3f} {e:9.3f} {e-s:7.3f}  q{r.get('Queue_Id','?')}  {name}  grid {r.get('Grid_Size','?')}")
