"""HBM bytes of the kernels of one rocprofv3 --pmc pass (FETCH_SIZE or WRITE_SIZE, KB), summed per kernel name and
divided by the number of pairs the run made:   python tools/pmc_pair.py <dir> <counter> <pairs>"""
import collections, csv, glob, sys
d, counter, pairs = sys.argv[1], sys.argv[2], float(sys.argv[3])
acc = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            k = r["Kernel_Name"].split("(")[0]
            acc[k] += float(r["Counter_Value"]); cnt[k] += 1
scale = (2.0 if counter == "FETCH_SIZE" else 1.0) * 1024.0 / pairs / 1e9     # gfx950: reads = 2 x FETCH_SIZE
for k in sorted(acc):
    if "k_linearise" in k or "k_factor" in k:
        print(f"{counter} {k}: {acc[k] * scale:.3f} GB per pair ({cnt[k]} dispatches)")
