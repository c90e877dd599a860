"""Where do a kernel's scratch (spill) instructions sit?  For each kernel of an assembly listing (tools/kernel_regs.sh leaves the last one in
/tmp/last_kernel.s) that spills, print the line range of its scratch_load / scratch_store instructions and the conditional branches that
jump INTO that range from before it - if every scratch instruction lies behind such a branch, the straight-line (hot) path executes none.
python tools/isa_scratch_blocks.py [listing.s] [name filter]"""
import re, sys
path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/last_kernel.s"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
s = open(path).read()
for name in dict.fromkeys(re.findall(r"^(_Z\S+):", s, re.M)):
    if flt not in name:
        continue
    i = s.index(name + ":")
    body = s[i:s.index(".Lfunc_end", i)].split("\n")
    sc = [n for n, l in enumerate(body) if "scratch_" in l]
    if not sc:
        continue
    labs = {m.group(1): n for n, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    first, last = sc[0], sc[-1]
    ends = [n for n, l in enumerate(body) if l.strip().startswith("s_endpgm") and n < first]
    if not ends:
        print(f"{name[:100]}\n   {len(sc)} scratch instructions in lines {first}..{last}, no s_endpgm in front of them: they are on the main path")
        continue
    E = ends[-1]
    hot_sc = [n for n in sc if n <= E]
    entries = [(n, l.split()[0], labs[l.split()[-1]]) for n, l in enumerate(body[:E + 1])
               if re.search(r"s_cbranch", l) and l.split()[-1] in labs and labs[l.split()[-1]] > E]
    print(f"{name[:100]}\n   {len(body)} lines; the main path ends with s_endpgm at line {E}; scratch instructions on it: {len(hot_sc)}; "
          f"behind it: {len(sc) - len(hot_sc)} (lines {first}..{last}), entered only through {entries}")
