"""(round 6) Which loops of a kernel wait for ALL their loads?  tools/isa_loop_waits.py <listing.s> [kernel substring]
For every backward branch (a loop) that contains vector-memory loads: the s_waitcnt vmcnt(N) inside it.  A software prefetch whose loads sit behind a
branch (`if (more) fetch(next)`) makes the wait-count pass merge two paths with different numbers of loads in flight; it then assumes nothing younger is
outstanding and emits vmcnt(0) (or a count far too small) where the source intended "all but the prefetch" - the prefetch distance collapses to less than
one trip (fdn_ffn_tail, round 6: profiles/r06_tail_mid_trace_before.txt).  Listing: hipcc ... --save-temps=obj (tools/kernel_regs.sh leaves /tmp/last_kernel.s)."""
import re, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+):\s*; @\1\n(.*?)s_endpgm', s, flags=re.M | re.S):
    name, body = m.group(1), m.group(2).split('\n')
    if flt not in name or '.num_vgpr' in name:
        continue
    labels = {}
    for n, l in enumerate(body):
        mm = re.match(r'^(\.LBB\d+_\d+):', l.strip())
        if mm:
            labels[mm.group(1)] = n
    loops = []
    for n, l in enumerate(body):
        mm = re.match(r'\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
        mb = re.match(r'\s*s_branch\s+(\.LBB\d+_\d+)', l)
        t = (mm or mb)
        if t and t.group(1) in labels and labels[t.group(1)] < n:
            loops.append((labels[t.group(1)], n))
    out = []
    for a, b in loops:
        seg = body[a:b]
        loads = sum(1 for l in seg if re.match(r'\s*(buffer_load|global_load|flat_load|scratch_load)', l))
        stores = sum(1 for l in seg if re.match(r'\s*(buffer_store|global_store|flat_store)', l))
        waits = [int(x) for l in seg for x in re.findall(r'vmcnt\((\d+)\)', l)]
        if loads and waits:
            out.append(f"   loop lines {a}-{b} ({b - a} lines): {loads} loads, {stores} stores, vmcnt waits {waits}")
    if out:
        print(re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', name)[:100])
        print("\n".join(out))
