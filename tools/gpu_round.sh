#!/bin/bash
# one GPU-box visit of a round: the gpu-marked tests, the rocprofv3 evidence of the default bench command (tools/profile_round.sh)
# and the per-config table.  usage (via gpurun, from the repo root): tools/gpu_round.sh r04 [notests]
TAG=$1
mkdir -p gpurun_out
if [ "${2:-}" != notests ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gpu_tests.txt 2>&1
  echo "gpu tests rc=$?"; tail -4 gpurun_out/${TAG}_gpu_tests.txt
fi
# multi-GPU path, every round: one real RCCL rank through the step's own all-reduce (nif_comm_init_rank + selftest) in
# bench.py, and the n = 1 forms of nif_comm_init_all / nif_train_step_multi + the 8-process start rehearsal (tests/test_gpu_distributed.py)
timeout 300 python bench.py --force-dist --steps 5 --warmup 2 --no-cpu-baseline --no-configs > gpurun_out/${TAG}_force_dist.json 2> gpurun_out/${TAG}_force_dist.err
echo "force-dist bench rc=$? $(python -c "import json;d=json.load(open('gpurun_out/${TAG}_force_dist.json'));print(d['dist']['rccl_ranks_seen'], d['ms_per_step'])" 2>&1)"
timeout 900 python -m pytest tests/test_gpu_distributed.py -m gpu -q > gpurun_out/${TAG}_gpu_distributed.txt 2>&1
echo "distributed rehearsal rc=$?"; tail -2 gpurun_out/${TAG}_gpu_distributed.txt
tools/profile_round.sh $TAG
timeout 900 python tools/bench_configs.py --out gpurun_out/${TAG}_configs_doc.json > gpurun_out/${TAG}_configs.json 2> gpurun_out/${TAG}_configs.err
echo "configs rc=$?"
cat gpurun_out/${TAG}_default/bench.json
