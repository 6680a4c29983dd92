#!/bin/bash
# rocprofv3 evidence of one round: kernel trace + stats, then PMC passes (separate runs with --kernel-trace only; FETCH_SIZE and
# WRITE_SIZE each in a pass of their own, as MI355X_MICROARCH.md prescribes) of the default bench.py command.
# usage (on the GPU box, from the repo root): tools/profile_round.sh r03
set -u
TAG=$1
R=$PWD
export TMPDIR=/tmp
ARGS="--steps 6 --warmup 3 --no-cpu-baseline --no-configs"
O=$R/gpurun_out/${TAG}_default
rm -rf $O; mkdir -p $O
python -c "import bench; print(bench.csrc_sha())" > $O/csrc_sha.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/bench.py $ARGS > $O/kt_bench.json 2> $O/kt_err.txt
echo "ktrace rc=$?"
i=0
for grp in "FETCH_SIZE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
           "WRITE_SIZE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/p$i -o p$i -- python $R/bench.py $ARGS > $O/p$i.json 2> $O/p$i.err
  echo "pmc pass $i rc=$?"
done
cd $R
python tools/pmc_summary.py $O > $O/pmc.md
f=$(find $O/kt -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $O/kernel_stats.csv
python bench.py > $O/bench.json 2> $O/bench.err
ls $O
