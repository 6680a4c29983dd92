#!/bin/bash
for f in 0 1; do
  r=$(NIF_FUSED_GW=$f python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('%.3f ms/step  snet %.3f gw %.3f pbw %.3f pfw %.3f red %.3f loss %s' % (d['ms_per_step'], k['snet'], k['gw'], k['pnet_bwd'], k['pnet_fwd'], k['reduce'], d['config']['final_loss']))")
  echo "fused_gw $f: $r"
done
