#!/bin/bash
# usage: tools/exp/abl_run.sh  -- runs bench.py quickly and prints ms/step + kernel groups (the lib must be prebuilt with the ablation flags)
python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('%.3f ms/step  snet %.3f gw %.3f pbw %.3f pfw %.3f' % (d['ms_per_step'], k['snet'], k['gw'], k['pnet_bwd'], k['pnet_fwd']))"
