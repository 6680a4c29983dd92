#!/bin/bash
# A/B of measurement builds (tools/build_variant.py): bench line + per-kernel times of each library given by tag
# usage: tools/exp/ab_libs.sh tag1 tag2 ...   ("main" = the product library)
for t in "$@"; do
  if [ "$t" = main ]; then L=$PWD/nif_amd/libnif_hip.so; else L=$PWD/nif_amd/libnif_hip_$t.so; fi
  NIF_LIB=$L python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/ab_$t.json 2> gpurun_out/ab_$t.err
  python - "$t" <<PY
import json, sys
t = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/ab_%s.json" % t).read().strip().split("\n")[-1])
    print(t, "ms/step %.3f" % d["ms_per_step"], {k: round(v, 3) for k, v in d["kernel_ms"].items() if v})
except Exception as e:
    print(t, "failed", e, open("gpurun_out/ab_%s.err" % t).read()[-600:])
PY
done
