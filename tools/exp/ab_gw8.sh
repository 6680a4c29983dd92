#!/bin/bash
# A/B of k_gw8 builds (tools/build_variant.py tags; "main" = product) on the configs whose gradients it owns
# usage: tools/exp/ab_gw8.sh tag ...     -> gpurun_out/ab_gw8.txt
out=gpurun_out/ab_gw8.txt; : > $out
for t in "$@"; do
  if [ "$t" = main ]; then L=$PWD/nif_amd/libnif_hip.so; else L=$PWD/nif_amd/libnif_hip_$t.so; fi
  for cfg in cfg3_ms_6x128_2d cfg4_linear_nif_3d; do
    NIF_LIB=$L python tools/bench_configs.py --only $cfg --steps 20 --warmup 5 2>&1 | grep "^$cfg" | python -c "
import sys, json
for line in sys.stdin:
    k, _, v = line.partition(' ')
    d = json.loads(v)
    print('$t', k, d['ms_per_step'], 'gw', d['kernel_ms']['gw'], 'snet', d['kernel_ms']['snet'])
" >> $out
  done
done
cat $out
