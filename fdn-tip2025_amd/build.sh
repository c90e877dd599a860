#!/bin/bash
# Build libfdn_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
#   OUT=path BUILD=dir EXTRA="flags" ./build.sh   builds a variant elsewhere (A/B experiments)
set -e
cd "$(dirname "$0")"
OUT=${OUT:-fdn_hip/libfdn_hip.so}
BUILD=${BUILD:-build}
mkdir -p fdn_hip "$BUILD"
# The SLP vectoriser pairs fp32 ops into v_pk_* at the price of v_mov shuffles and ~45 more VGPRs; the
# VALU-issue-bound kernels listed here measure faster without it (fdffn_mid 2.35 -> 1.89 ms at level 1).
NOSLP="patchfft ffn_tail fdsa_full"
OBJS=""
PIDS=""
NEWEST_HDR=$(ls -t csrc/*.hpp csrc/*.inc ../include/fdn_hip.h build.sh | head -1)
for f in csrc/*.hip; do
  n=$(basename "${f%.hip}")
  o=$BUILD/$n.o
  fl=""
  for k in $NOSLP; do [ "$k" = "$n" ] && fl="-fno-slp-vectorize"; done
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$NEWEST_HDR" -nt "$o" ]; then
    rm -f "$o"                      # a failed compile must not leave a stale object for the link below
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $fl $EXTRA -c "$f" -o "$o" &
    PIDS="$PIDS $!"
  fi
  OBJS="$OBJS $o"
done
for p in $PIDS; do wait "$p" || { echo "build.sh: a hipcc job failed" >&2; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" $OBJS
echo "built $OUT"
