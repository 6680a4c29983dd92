#!/bin/bash
# per-config kernel trace: rocprofv3 --kernel-trace --stats of tools/bench_configs.py --only <cfg> for every BASELINE-named
# config beyond the headline -> gpurun_out/ktc/<cfg>_kernel_stats.csv (summarised by tools/ktrace_configs_md.py into profiles/)
R=$PWD
export TMPDIR=/tmp
rm -rf $R/gpurun_out/ktc; mkdir -p $R/gpurun_out/ktc
cd /tmp
for cfg in "$@"; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktc_$cfg -o kt -- python $R/tools/bench_configs.py --only $cfg --exact --steps 10 --warmup 3 > $R/gpurun_out/ktc/$cfg.out 2> $R/gpurun_out/ktc/$cfg.err
  f=$(find /tmp/ktc_$cfg -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $R/gpurun_out/ktc/${cfg}_kernel_stats.csv
done
ls $R/gpurun_out/ktc
