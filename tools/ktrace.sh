#!/bin/bash
R=$PWD
rm -rf $R/gpurun_out/ktrace; mkdir -p $R/gpurun_out/ktrace
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ktrace -o kt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs > $R/gpurun_out/ktrace/bench.json 2> $R/gpurun_out/ktrace/err.txt
echo rc=$?
ls $R/gpurun_out/ktrace
