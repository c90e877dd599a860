#!/bin/bash
# rocprofv3 passes of one round over tools/profile_forward.py (run ON THE GPU BOX from the repo root):
#   tools/profile_passes.sh <outdir under gpurun_out> [profile_forward.py args]
# Kernel trace and every counter group in its own run (--pmc is never combined with another trace domain); the program after `--`
# is python itself.  TCC counters: the memory-side read requests by size (32 / 64 / 128 B) give the read bytes exactly; FETCH_SIZE
# and WRITE_SIZE are collected as MI355X_MICROARCH.md prescribes (separate passes) for the cross-check.
set -u
OUT=$1; shift
R=$(pwd)
mkdir -p "$R/$OUT"
cd /tmp && export TMPDIR=/tmp
# (each pass under its own timeout: a counter group rocprofv3 cannot collect aborts the tool and would otherwise hang until the box's limit)
run() { name=$1; shift; FDN_CALL_LOG="$R/$OUT/calls_$name.json" timeout 420 rocprofv3 "$@" --output-format csv -d "$R/$OUT/$name" -o p -- python3 "$R/tools/profile_forward.py" $ARGS > "$R/$OUT/$name.log" 2>&1; }
ARGS="$*"
run trace --kernel-trace --stats
run rd --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run wr --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run sq --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
cd "$R"
