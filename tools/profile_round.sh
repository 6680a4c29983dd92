#!/bin/bash
# rocprofv3 evidence of one round: kernel trace + stats and PMC passes (separate runs, --kernel-trace only) of bench.py,
# for the default step and for the opt-in fused weight-gradient kernel (NIF_FUSED_GW=1).
# usage (on the GPU box, from the repo root): tools/profile_round.sh r02
set -u
TAG=$1
R=$PWD
export TMPDIR=/tmp
ARGS="--steps 6 --warmup 3 --no-cpu-baseline --no-extras"
for variant in default fused; do
  if [ $variant = fused ]; then export NIF_FUSED_GW=1; else unset NIF_FUSED_GW; fi
  O=$R/gpurun_out/${TAG}_$variant
  rm -rf $O; mkdir -p $O
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/bench.py $ARGS > $O/kt_bench.json 2> $O/kt_err.txt
  echo "$variant ktrace rc=$?"
  i=0
  for grp in "FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" \
             "TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/p$i -o p$i -- python $R/bench.py $ARGS > $O/p$i.json 2> $O/p$i.err
    echo "$variant pmc pass $i rc=$?"
  done
  cd $R
  python tools/pmc_summary.py $O > $O/pmc.md
  f=$(find $O/kt -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/kernel_stats.csv
done
ls $R/gpurun_out/${TAG}_default $R/gpurun_out/${TAG}_fused
