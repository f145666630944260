"""Per-kernel averages of arbitrary rocprofv3 PMC counters (one --pmc pass):
    python tools/pmc_sq.py <dir> [kernel substring ...]"""
import csv, glob, sys, collections
d = sys.argv[1]
want = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(cfn::Params")[0].replace("void ", "")
        if want and not any(w in k for w in want):
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        tail = v[len(v) // 2:]
        print(f"   {c:28s} {sum(tail) / len(tail):16.1f}   ({len(v)} dispatches)")
