"""Summarise a rocprofv3 --pmc pass: tools/pmc_kernel.py <dir> <kernel substring> [...]: per kernel (name, grid) the summed counters."""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if any(s in k for s in sys.argv[2:]):
            key = k.replace("void (anonymous namespace)::", "")[:70] + " grid=" + r["Grid_Size"]
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(key, r["Counter_Name"])] += 1
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"    {c:28s} {x / cnt[(k, c)]:.5g}  (per dispatch, {cnt[(k, c)]} dispatches)")
