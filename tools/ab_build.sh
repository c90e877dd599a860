#!/bin/bash
# tools/ab_build.sh <name> <source stem> "<flags>": a variant of libfdn_hip.so with ONE source recompiled under extra flags
# (A/B builds for tools/ab_libs.py).  Writes abx/lib_<name>.so (travels to the GPU box; git-ignored).
set -e
cd "$(dirname "$0")/../fdn-tip2025_amd"
name=$1; stem=$2; flags=$3
mkdir -p ../abx ../abtest/b_$name
cp -u build/*.o ../abtest/b_$name/
rm -f ../abtest/b_$name/$stem.o
OUT=../abx/lib_$name.so BUILD=../abtest/b_$name EXTRA="$flags" ./build.sh | tail -1
