#!/bin/bash
# rocprofv3 counter passes over tools/pmc_fdsa_full.py (run ON THE GPU BOX from the repo root): tools/pmc_fdsa_full.sh <outdir under gpurun_out>
set -u
OUT=$1; R=$(pwd); mkdir -p "$R/$OUT"
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 "$@" --output-format csv -d "$R/$OUT/$name" -o p -- python3 "$R/tools/pmc_fdsa_full.py" 3 > "$R/$OUT/$name.log" 2>&1; }
run trace --kernel-trace --stats
run sq1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES
run sq2 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM
run sq3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_LEVEL_LDS
run sq4 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_WAVES_EQ_64 SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
cd "$R"
for p in sq1 sq2 sq3 sq4; do echo "== $p"; python3 tools/pmc_report.py $OUT/$p fdsa_ 2>&1 | head -40; done > $OUT/report.txt
grep -h "fdsa" $OUT/trace/*kernel_stats.csv 2>/dev/null | head -8 >> $OUT/report.txt
cat $OUT/report.txt
