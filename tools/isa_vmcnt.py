"""Per kernel: MFMA count and the s_waitcnt vmcnt(..) that sit inside MFMA runs (a store/load wait in the middle of the
matrix pipe's work).  tools/isa_vmcnt.py file.s [filter]"""
import re, sys
s = open(sys.argv[1]).read(); flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", s, re.S | re.M):
    name, body = m.group(1), m.group(2).split("\n")
    if flt not in name: continue
    ins = [l.strip() for l in body if l.strip() and not l.strip().startswith((";", "."))]
    mf = [i for i, l in enumerate(ins) if l.startswith("v_mfma")]
    if not mf: continue
    inside = []
    for i, l in enumerate(ins):
        if l.startswith("s_waitcnt") and "vmcnt" in l:
            near = any(x.startswith("v_mfma") for x in ins[max(0, i - 4):i]) and any(x.startswith("v_mfma") for x in ins[i + 1:i + 5])
            if near: inside.append(re.search(r"vmcnt\((\d+)\)", l).group(1))
    print(f"{name[:90]:90s} mfma={len(mf):4d} vmcnt-inside={len(inside):3d} {inside[:24]}")
