"""Which launches sit next to the __amd_rocclr_copyBuffer dispatches of a forward?  Reads the kernel-trace CSV of
    rocprofv3 --kernel-trace --output-format csv -d <dir> -o p -- python3 tools/profile_forward.py
and prints, for the copy dispatches of the LAST forward, the histogram of (previous kernel, next kernel).   python tools/copy_sources.py <dir>"""
import collections, csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
short = lambda n: n.split("(")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:60]
idx = [i for i, n in enumerate(names) if "copyBuffer" in n]
print(len(idx), "copy dispatches of", len(names))
half = [i for i in idx if i > len(names) // 2]
h = collections.Counter((short(names[i - 1]), short(names[i + 1]) if i + 1 < len(names) else "-") for i in half)
for (a, b), c in h.most_common(25):
    print(f"{c:4d}  after {a:60s} before {b}")
print("(copies in the first half of the trace belong to the warm-up: weight packing, FFT tables)")
