#!/bin/bash
# Build libfdn_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=fdn_hip/libfdn_hip.so
mkdir -p fdn_hip build
OBJS=""
for f in csrc/*.hip; do
  o=build/$(basename "${f%.hip}").o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ csrc/common.hpp -nt "$o" ] || [ ../include/fdn_hip.h -nt "$o" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -c "$f" -o "$o" &
  fi
  OBJS="$OBJS $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" $OBJS
echo "built $OUT"
