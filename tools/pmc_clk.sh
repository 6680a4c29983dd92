#!/bin/bash
R=$PWD
mkdir -p $R/gpurun_out/pmc_clk
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $R/gpurun_out/pmc_clk/p1 -o p1 -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-configs > $R/gpurun_out/pmc_clk/p1.json 2> $R/gpurun_out/pmc_clk/p1.err
echo rc=$?
find $R/gpurun_out/pmc_clk -name "*.csv" | head
