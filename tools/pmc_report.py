"""Average the per-dispatch counters of a rocprofv3 --pmc csv by (kernel, grid): tools/pmc_report.py <dir> [name filter]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    if flt not in r["Kernel_Name"]:
        continue
    k = r["Kernel_Name"][:70] + " grid=" + r["Grid_Size"] + " lds=" + r.get("LDS_Block_Size", "?") + " vgpr=" + r.get("VGPR_Count", "?")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, v in agg.items():
    print(k)
    print("    ", {c: round(x / cnt[(k, c)]) for c, x in v.items()})
