#!/bin/bash
# Collect PMC counters for the bench in separate passes (rocprofv3; one counter group per run).
# usage: tools/pmc_run.sh <outdir-under-gpurun_out> [bench args...]
set -u
OUT=$1; shift
R=$PWD
mkdir -p $R/gpurun_out/$OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "FETCH_SIZE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
           "WRITE_SIZE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/$OUT/p$i -o p$i -- python $R/bench.py "$@" > $R/gpurun_out/$OUT/p$i.json 2> $R/gpurun_out/$OUT/p$i.err
  echo "pass $i rc=$?"
done
find $R/gpurun_out/$OUT -name "*.csv" | head -20
