#!/bin/bash
# HBM traffic of the dominant kernels of the BASELINE-named configs beyond the headline: rocprofv3 --pmc with FETCH_SIZE and
# WRITE_SIZE each in a pass of its own (--kernel-trace only, as MI355X_MICROARCH.md prescribes) around tools/bench_configs.py
# --only <cfg> --exact; per-kernel means by tools/pmc_summary.py -> gpurun_out/pmcc/<cfg>.md (tools/pmc_configs_md.py -> profiles/)
R=$PWD
export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmcc; mkdir -p $R/gpurun_out/pmcc
(cd $R && python -c "import bench; print(bench.csrc_sha())") > $R/gpurun_out/pmcc/csrc_sha.txt
cd /tmp
for cfg in "$@"; do
  O=/tmp/pmcc_$cfg; rm -rf $O; mkdir -p $O
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/p$i -o p$i -- python $R/tools/bench_configs.py --only $cfg --exact --steps 6 --warmup 3 > $O/p$i.log 2> $O/p$i.err
  done
  python $R/tools/pmc_summary.py $O > $R/gpurun_out/pmcc/$cfg.md
  grep "^$cfg " $O/p1.log > $R/gpurun_out/pmcc/$cfg.out
done
ls $R/gpurun_out/pmcc
