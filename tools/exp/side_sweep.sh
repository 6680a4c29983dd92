#!/bin/bash
for side in 0 1; do for rep in 1 2; do
  r=$(NIF_SIDE_PNET=$side python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('%.3f ms/step  snet %.3f gw %.3f pbw %.3f pfw %.3f red %.3f' % (d['ms_per_step'], k['snet'], k['gw'], k['pnet_bwd'], k['pnet_fwd'], k['reduce']))")
  echo "side_pnet $side: $r"
done; done
