"""Waterfall loops (v_readfirstlane + v_cmp_eq + s_and_saveexec ... s_cbranch_execnz) per kernel of a hipcc -S listing: tools/isa_waterfall.py file.s
The compiler wraps a buffer access in one when an operand that must be scalar (soffset, descriptor) sits in a vector register - a
strength-reduced loop offset it chose to keep in a VGPR (fdffn_mid, round 4), a descriptor that really differs per lane (rfft_rows_ln);
__builtin_amdgcn_readfirstlane on the offset / one descriptor with the varying part in the per-lane offset removes them."""
import re, sys


def waterfalls(path):
    """{kernel symbol: (waterfall loops, v_readfirstlane count)}"""
    lines = open(path).read().split('\n')
    cur, stats = None, {}
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            cur = m.group(1); stats[cur] = [0, 0]; continue
        if cur is None:
            continue
        t = l.strip()
        if t.startswith('s_endpgm'):
            cur = None; continue
        if t.startswith('s_cbranch_execnz'):
            win = ' '.join(lines[max(0, i - 14):i])            # the loop body (a few lines up) holds v_readfirstlane + v_cmp_eq
            if 'v_readfirstlane' in win and 'v_cmp_eq' in win:
                stats[cur][0] += 1
        if t.startswith('v_readfirstlane'):
            stats[cur][1] += 1
    return {k: tuple(v) for k, v in stats.items()}


if __name__ == "__main__":
    for k, (w, r) in waterfalls(sys.argv[1]).items():
        if w or r > 8:
            print(f"{k[:100]:100s} waterfalls {w:3d}  readfirstlanes {r:3d}")
