#!/bin/bash
# HBM traffic counters (FETCH_SIZE / WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes) of bench.py for the
# default step and the fused weight-gradient kernel.  usage: tools/profile_traffic.sh r02
set -u
TAG=$1
R=$PWD
export TMPDIR=/tmp
ARGS="--steps 6 --warmup 3 --no-cpu-baseline --no-extras"
for variant in default fused; do
  if [ $variant = fused ]; then export NIF_FUSED_GW=1; else unset NIF_FUSED_GW; fi
  O=$R/gpurun_out/${TAG}_$variant
  mkdir -p $O
  cd /tmp
  i=3
  for grp in "FETCH_SIZE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "WRITE_SIZE SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
    i=$((i+1))
    rm -rf $O/p$i
    timeout 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/p$i -o p$i -- python $R/bench.py $ARGS > $O/p$i.json 2> $O/p$i.err
    echo "$variant pmc pass $i rc=$?"
  done
  cd $R
  rm -rf $O/p1
  python tools/pmc_summary.py $O > $O/pmc.md
done
