#!/bin/bash
# rocprofv3 evidence of the Sobolev step (BASELINE configs[4], tools/bench_configs.py cfg5_sobolev_2d_4x64*): kernel trace + stats,
# then FETCH_SIZE and WRITE_SIZE in a counter pass each.  usage (on the GPU box, from the repo root): tools/profile_sobolev.sh r03
set -u
TAG=$1
R=$PWD
export TMPDIR=/tmp
CMD="python $R/tools/bench_configs.py --only ${SOB_ONLY:-cfg5_sobolev_2d_4x64} --steps 6 --warmup 3"
O=$R/gpurun_out/${TAG}_sobolev
rm -rf $O; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $CMD > $O/kt.log 2> $O/kt_err.txt
echo "ktrace rc=$?"
i=0
for grp in "FETCH_SIZE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
           "WRITE_SIZE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/p$i -o p$i -- $CMD > $O/p$i.log 2> $O/p$i.err
  echo "pmc pass $i rc=$?"
done
cd $R
python tools/pmc_summary.py $O > $O/pmc.md
f=$(find $O/kt -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $O/kernel_stats.csv
ls $O
