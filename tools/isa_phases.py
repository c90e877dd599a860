"""Static instruction mix per phase of fdsa_full_kernel: compiles a copy of csrc/fdsa_full.hip with assembler comments at the
phase headers (`// ---- Pn`) and counts VALU / LDS / MFMA / VMEM / SALU instructions between them.
python tools/isa_phases.py [kernel mangled-name substring, default Li32ELi2ELb1] [extra hipcc flags]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sel = sys.argv[1] if len(sys.argv) > 1 else "Li32ELi2ELb1"
src = open(os.path.join(ROOT, "fdn-tip2025_amd/csrc/fdsa_full.hip")).read()
out = []
for ln in src.split("\n"):
    m = re.match(r"\s*// ---- (P\d|epilogue)", ln)
    if m:
        out.append('asm volatile("; MARK_%s" ::: "memory");' % m.group(1))
    if ln.strip().startswith("// statistics of the three groups"):
        out.append('asm volatile("; MARK_combine" ::: "memory");')
    out.append(ln)
d = tempfile.mkdtemp()
open(os.path.join(d, "k.hip"), "w").write("\n".join(out))
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-fno-slp-vectorize",
                "-I" + os.path.join(ROOT, "fdn-tip2025_amd/csrc")] + sys.argv[2:] + ["-c", os.path.join(d, "k.hip"), "-o", os.path.join(d, "k.o"),
                "--save-temps=obj"], check=True, stderr=subprocess.DEVNULL)
s = open(os.path.join(d, "k-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
name = [m for m in re.findall(r"^(_Z\S*fdsa_full_kernel\S*):", s, re.M) if sel in m][0]
body = s[s.index(name + ":"):]
body = body[:body.index(".Lfunc_end")].split("\n")
cur, c, order = "prologue", {}, []
for l in body:
    m = re.search(r"; MARK_(\w+)", l)
    if m:
        cur = m.group(1)
    t = l.strip().split(" ")[0]
    if not t or t[0] in ".;" or t.endswith(":"):
        continue
    k = "mfma" if "mfma" in t else "valu" if t.startswith("v_") else "lds" if t.startswith("ds_") else "vmem" if t.startswith(("buffer_", "global_")) else "salu"
    if cur not in c:
        c[cur] = {}
        order.append(cur)
    c[cur][k] = c[cur].get(k, 0) + 1
print(name)
for k in order:
    print(f"  {k:10s}", "  ".join(f"{a} {c[k].get(a, 0):4d}" for a in ("valu", "lds", "mfma", "vmem", "salu")))
loop = [k for k in order if k.startswith("P")]
print("  per chunk  valu", sum(c[k].get("valu", 0) for k in loop), " lds", sum(c[k].get("lds", 0) for k in loop), " mfma", sum(c[k].get("mfma", 0) for k in loop))
