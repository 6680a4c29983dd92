#!/bin/bash
# sweep of the two-stream chunk pipeline: chunk size x fused-kernel workgroups per chunk (bench.py, 2^20 points)
for cfg in "0 768" "262144 512" "131072 512" "131072 640" "131072 768" "65536 512" "65536 384" "32768 512" "32768 256"; do
  set -- $cfg
  r=$(NIF_PIPE_CHUNK=$1 NIF_PIPE_WGS=$2 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('%.3f ms/step  snet %.3f gw %.3f pbw %.3f pfw %.3f red %.3f' % (d['ms_per_step'], k['snet'], k['gw'], k['pnet_bwd'], k['pnet_fwd'], k['reduce']))")
  echo "chunk $1 wgs $2: $r"
done
