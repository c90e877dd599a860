"""Merge the rocprofv3 passes of tools/profile_passes.sh into the per-kernel summary of a round:
    python tools/profile_merge.py gpurun_out/<dir> profiles/r02_x            -> profiles/r02_x_traffic_groups.json, _kernel_stats.csv, _summary.txt

Alignment: the logged forward is the LAST one of each pass, every C-ABI call launches exactly one kernel, so the last N library
kernels of a pass are the N logged calls in order (checked: the kernel-name sequence must agree between passes).
HBM read bytes per dispatch = 32 n32 + 64 n64 + 128 n128 from the memory-side request counters (TCC_EA0_RDREQ_*; their sum must
equal TCC_EA0_RDREQ); write bytes = 64 n64 + 32 (n - n64).  FETCH_SIZE / WRITE_SIZE (KB) are kept beside them: on gfx950 FETCH_SIZE
tallies a 128-B request as 64 B, so fetch_over_exact shows per kernel which correction it needs (0.5 = all 128-B requests)."""
import csv, glob, json, os, sys, collections

src, dst = sys.argv[1], sys.argv[2]
MINE = ("(anonymous namespace)", "pack_guidance_kernel")


def short(k):
    return k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]


def counters(name):
    f = glob.glob(f"{src}/{name}/**/*counter_collection.csv", recursive=True)
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        if not any(m in r["Kernel_Name"] for m in MINE):
            continue
        d = rows.setdefault(int(r["Dispatch_Id"]), {"kernel": short(r["Kernel_Name"]), "grid": int(r["Grid_Size"])})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [rows[k] for k in sorted(rows)]


def trace():
    f = glob.glob(f"{src}/trace/**/*kernel_trace.csv", recursive=True)
    rows = []
    for r in csv.DictReader(open(f[0])):
        if any(m in r["Kernel_Name"] for m in MINE):
            rows.append({"kernel": short(r["Kernel_Name"]), "id": int(r["Dispatch_Id"]), "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                         "vgpr": r.get("VGPR_Count") or r.get("Arch_VGPR_Count"), "lds": r.get("LDS_Block_Size")})
    rows.sort(key=lambda r: r["id"])
    return rows


calls = json.load(open(f"{src}/calls_trace.json"))["calls"]
N = len(calls)
tr = trace()[-N:]
passes = {n: counters(n)[-N:] for n in ("rd", "wr", "fetch", "write", "sq")}
for n, rows in passes.items():
    assert len(rows) == N and [r["kernel"] for r in rows] == [t["kernel"] for t in tr], f"pass {n} does not align with the trace"
groups = collections.OrderedDict()
for i, c in enumerate(calls):
    gkey = c["group"] if "[" in c["group"] else f'{c["group"]}[{tr[i]["kernel"]}]'      # entry points without a shape key: split by kernel
    g = groups.setdefault(gkey, {"entry": c["entry"], "kernel": tr[i]["kernel"], "launches": 0, "ms": 0.0, "alg_flops": 0.0, "alg_bytes": 0.0,
                                      "rd": [0, 0, 0, 0], "wr": [0, 0], "fetch_kb": 0.0, "write_kb": 0.0, "valu": 0.0, "mfma_busy": 0.0, "lds_insts": 0.0,
                                      "waves": 0.0, "wait_inst": 0.0, "wave_cycles": 0.0})
    assert g["kernel"] == tr[i]["kernel"], (gkey, g["kernel"], tr[i]["kernel"])
    g["launches"] += 1
    g["ms"] += tr[i]["ns"] / 1e6
    g["alg_flops"] += c["flops"]
    g["alg_bytes"] += c["bytes"]
    r = passes["rd"][i]
    for j, k in enumerate(("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum")):
        g["rd"][j] += r.get(k, 0.0)
    w = passes["wr"][i]
    g["wr"][0] += w.get("TCC_EA0_WRREQ_sum", 0.0)
    g["wr"][1] += w.get("TCC_EA0_WRREQ_64B_sum", 0.0)
    g["fetch_kb"] += passes["fetch"][i].get("FETCH_SIZE", 0.0)
    g["write_kb"] += passes["write"][i].get("WRITE_SIZE", 0.0)
    s = passes["sq"][i]
    g["valu"] += s.get("SQ_INSTS_VALU", 0.0)
    g["mfma_busy"] += s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    g["lds_insts"] += s.get("SQ_INSTS_LDS", 0.0)
    g["waves"] += s.get("SQ_WAVES", 0.0)
    g["wait_inst"] += s.get("SQ_WAIT_INST_ANY", 0.0)
    g["wave_cycles"] += s.get("SQ_WAVE_CYCLES", 0.0)

_meta = json.load(open(f"{src}/calls_trace.json"))
def _code_commit():
    """the commit the profiled code was built from: FDN_PROFILE_COMMIT, or the file the builder writes before a gpurun call (the GPU box has no .git)"""
    c = os.environ.get("FDN_PROFILE_COMMIT")
    f = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "HEAD_AT_PROFILE")
    return c or (open(f).read().strip() if os.path.isfile(f) else None)


NSIMD, out = 1024.0, {"note": __doc__, "shape": _meta["shape"], "dtype": _meta.get("dtype", "f32"), "commit": _code_commit(), "groups": {}}
tot = {"ms": 0.0, "rd": 0.0, "wr": 0.0, "alg": 0.0, "valu_ms": 0.0, "mfma_ms": 0.0}
lines = []
for key, g in sorted(groups.items(), key=lambda kv: -kv[1]["ms"]):
    n = g["launches"]
    rd_total, n32, n64, n128 = g["rd"]
    rd_bytes = 32.0 * n32 + 64.0 * n64 + 128.0 * n128
    wr_bytes = 64.0 * g["wr"][1] + 32.0 * (g["wr"][0] - g["wr"][1])
    hbm = rd_bytes + wr_bytes
    sec = g["ms"] * 1e-3
    clk = 2.1e9                                                   # effective clock under load (MI355X_MICROARCH.md, DVFS)
    valu_ms = g["valu"] * 4.0 / NSIMD / clk * 1e3                  # 4 issue cycles per wave64 vector instruction (tools/micro/mfma_valu_coexec.hip)
    mfma_ms = g["mfma_busy"] / NSIMD / clk * 1e3
    out["groups"][key] = {
        "entry": g["entry"], "kernel": g["kernel"], "launches": n, "avg_ms": g["ms"] / n,
        "hbm_bytes_per_launch": hbm / n, "read_bytes_per_launch": rd_bytes / n, "write_bytes_per_launch": wr_bytes / n,
        "algorithmic_bytes_per_launch": g["alg_bytes"] / n, "traffic_over_algorithmic": hbm / g["alg_bytes"] if g["alg_bytes"] else None,
        "rdreq_by_size": {"32B": n32 / n, "64B": n64 / n, "128B": n128 / n, "sum_check": (n32 + n64 + n128) / rd_total if rd_total else None},
        "fetch_size_bytes_per_launch": g["fetch_kb"] * 1024.0 / n, "fetch_over_exact": g["fetch_kb"] * 1024.0 / rd_bytes if rd_bytes else None,
        "write_size_bytes_per_launch": g["write_kb"] * 1024.0 / n, "write_size_over_exact": g["write_kb"] * 1024.0 / wr_bytes if wr_bytes else None,
        "hbm_gbs": hbm / sec / 1e9, "alg_tflops": g["alg_flops"] / sec / 1e12,
        "valu_insts_per_wave": g["valu"] / g["waves"] if g["waves"] else None,
        "simd_time_frac": {"valu": valu_ms / g["ms"], "mfma": mfma_ms / g["ms"]},
    }
    tot["ms"] += g["ms"]; tot["rd"] += rd_bytes; tot["wr"] += wr_bytes; tot["alg"] += g["alg_bytes"]; tot["valu_ms"] += valu_ms; tot["mfma_ms"] += mfma_ms
    lines.append(f"{key[:58]:58s} n={n:4d} {g['ms']:8.2f} ms  hbm {hbm/1e9:7.2f} GB ({hbm/g['alg_bytes'] if g['alg_bytes'] else 0:4.2f}x alg) "
                 f"{hbm/sec/1e12:5.2f} TB/s  fetch/exact {g['fetch_kb']*1024/rd_bytes if rd_bytes else 0:4.2f}  valu {100*valu_ms/g['ms']:3.0f}% mfma {100*mfma_ms/g['ms']:3.0f}%")
out["total"] = {"kernel_ms": tot["ms"], "read_GB": tot["rd"] / 1e9, "write_GB": tot["wr"] / 1e9, "algorithmic_GB": tot["alg"] / 1e9,
                "valu_simd_ms": tot["valu_ms"], "mfma_simd_ms": tot["mfma_ms"]}
json.dump(out, open(dst + "_traffic_groups.json", "w"), indent=1)
with open(dst + "_summary.txt", "w") as f:
    f.write(f"single-stream forward, sum of library kernels {tot['ms']:.1f} ms; HBM read {tot['rd']/1e9:.1f} GB + write {tot['wr']/1e9:.1f} GB "
            f"(algorithmic {tot['alg']/1e9:.1f} GB); SIMD time if nothing stalled: vector ALU {tot['valu_ms']:.1f} ms + MFMA (fp32 and bf16 pipes) {tot['mfma_ms']:.1f} ms\n")
    f.write("\n".join(lines) + "\n")
st = glob.glob(f"{src}/trace/**/*kernel_stats.csv", recursive=True)
if st:
    open(dst + "_kernel_stats.csv", "w").write(open(st[0]).read())
print(open(dst + "_summary.txt").read()[:6000])
