#!/bin/bash
# one GPU-box visit of a round: the gpu-marked tests, the rocprofv3 evidence of the default bench command (tools/profile_round.sh)
# and the per-config table.  usage (via gpurun, from the repo root): tools/gpu_round.sh r04 [notests]
TAG=$1
mkdir -p gpurun_out
if [ "${2:-}" != notests ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gpu_tests.txt 2>&1
  echo "gpu tests rc=$?"; tail -4 gpurun_out/${TAG}_gpu_tests.txt
fi
tools/profile_round.sh $TAG
timeout 900 python tools/bench_configs.py --out gpurun_out/${TAG}_configs_doc.json > gpurun_out/${TAG}_configs.json 2> gpurun_out/${TAG}_configs.err
echo "configs rc=$?"
cat gpurun_out/${TAG}_default/bench.json
