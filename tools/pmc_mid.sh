#!/bin/bash
# Stall counters of fdn_fdffn_mid in fp32 and bf16 storage (run ON THE GPU BOX from the repo root): tools/pmc_mid.sh <outdir under gpurun_out>
set -u
OUT=$1; R=$(pwd); mkdir -p "$R/$OUT"
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 "$@" --output-format csv -d "$R/$OUT/$name" -o p -- python3 "$R/tools/pmc_mid.py" > "$R/$OUT/$name.log" 2>&1; }
run trace --kernel-trace --stats
run sq1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVES
run sq2 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM
run sq3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU
run tcp --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum
run tcc --pmc TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum
cd "$R"
for p in sq1 sq2 sq3 tcp tcc; do echo "== $p"; python3 tools/pmc_report.py $OUT/$p fdffn_mid 2>&1 | head -12; done > $OUT/report.txt
grep -h "fdffn_mid" $OUT/trace/*kernel_stats.csv 2>/dev/null | head -4 >> $OUT/report.txt
cat $OUT/report.txt
