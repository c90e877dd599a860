#!/bin/bash
# Registers / spills / LDS of every kernel of one csrc file:  tools/kernel_regs.sh <stem> [extra hipcc flags]   (e.g. fdsa_full -DFDN_FULL_AWA=0)
stem=$1; shift
d=$(mktemp -d)
fl=""; case " patchfft ffn_tail fdsa_full " in *" $stem "*) fl="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $fl "$@" -c "$(dirname "$0")/../fdn-tip2025_amd/csrc/$stem.hip" -o $d/o.o --save-temps=obj 2>/dev/null
python3 - "$d"/*gfx950.s <<'PY'
import re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r'\.group_segment_fixed_size:\s*(\d+).*?\.name:\s*(\S+).*?\.private_segment_fixed_size:\s*(\d+).*?\.sgpr_count:\s*(\d+).*?\.sgpr_spill_count:\s*(\d+).*?\.vgpr_count:\s*(\d+).*?\.vgpr_spill_count:\s*(\d+)', s, re.S):
    lds, name, scratch, sg, sgs, vg, vgs = m.groups()
    print(f"{name[:90]:90s} vgpr {vg:>3s} spill {vgs:>3s}  sgpr {sg:>3s} spill {sgs:>3s}  scratch {scratch:>5s}  lds {lds}")
PY
cp "$d"/*gfx950.s /tmp/last_kernel.s
rm -rf "$d"
