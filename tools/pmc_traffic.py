"""Merge rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (separate runs, same command) into a per-kernel
traffic summary: tools/pmc_traffic.py <fetch_dir> <write_dir> <kernel_trace_dir> <out.json> [calib_fetch_dir]

FETCH_SIZE / WRITE_SIZE are reported in KB.  The optional calibration directory holds a FETCH_SIZE pass over
`tools/bench_kernels.py stats` (chan_stats reads exactly B*C*P*4 bytes with the same 4-byte-per-lane coalesced
loads as the hot kernels): its measured/expected ratio is stored as `fetch_calibration`."""
import csv, glob, json, sys, collections

def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"]) * 1024.0
    return agg

def durations(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        agg[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    return agg

fetch, write, dur = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE"), durations(sys.argv[3])
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over the same bench.py command; KB*1024; "
               "fetch_GB_raw is the counter as reported (MI355X_MICROARCH.md: gfx950 tallies a 128-B request as 64 B for wide "
               "coalesced reads: bench.py doubles it); ms from the --kernel-trace pass",
       "kernels": {}}
for k in sorted(fetch, key=lambda k: -dur.get(k, 0)):
    out["kernels"][k] = {"launches": fetch[k][0], "fetch_GB_raw": fetch[k][1] / 1e9, "write_GB": write.get(k, [0, 0.0])[1] / 1e9,
                         "ms": dur.get(k, 0.0)}
if len(sys.argv) > 5:
    # first chan_stats launch of tools/bench_kernels.py stats: level 1, C = 32 channels, one thread per (b, pixel):
    # it reads exactly Grid_Size * 32 * 4 bytes with 4-byte-per-lane coalesced loads
    f = glob.glob(sys.argv[5] + "/**/*counter_collection.csv", recursive=True)[0]
    row = [r for r in csv.DictReader(open(f)) if "chan_stats" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"][0]
    exp = int(row["Grid_Size"]) * 32 * 4 * (4 if "chan_stats4" in row["Kernel_Name"] else 1)      # float4 form: 4 pixels per thread
    raw = float(row["Counter_Value"]) * 1024.0
    out["fetch_calibration"] = {"kernel": row["Kernel_Name"].split("(")[-2].split("::")[-1] + " (B=8, C=32, 736x1280)", "fetch_bytes_raw": raw, "expected_bytes": exp,
                                "raw_over_expected": raw / exp,
                                "note": "4-byte-per-lane coalesced loads: FETCH_SIZE reports half of the bytes, the same factor "
                                        "MI355X_MICROARCH.md gives for 16-byte lanes, so bench.py doubles fetch_GB_raw"}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps({k: v for k, v in list(out["kernels"].items())[:6]}, indent=1))
