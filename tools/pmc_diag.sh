#!/bin/bash
# memory-path PMC passes of the bench's fused kernel: L2 write-back / EA stalls, TA / TCP stalls, request latencies.
# usage: tools/pmc_diag.sh <tag> [NIF_LIB path]
set -u
TAG=$1
R=$PWD
[ $# -ge 2 ] && export NIF_LIB=$2
export TMPDIR=/tmp
O=$R/gpurun_out/diag_$TAG
rm -rf $O; mkdir -p $O
cd /tmp
i=0
# (r3: the passes with TCC_EA0_WRREQ* / TA_* / TCP_UTCL1_* counters did not finish within 300 s on this pool -- rocprofv3 hung -- and
#  cost 30 GPU-minutes; this is the pass that works)
for grp in "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_IB_STALL_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/p$i -o p$i -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras > $O/p$i.json 2> $O/p$i.err
  echo "pass $i rc=$?"
done
cd $R
python tools/pmc_summary.py $O | awk '/k_snet4/{f=1} f{print} /^$/{if(f)exit}' > $O/snet.md
cat $O/snet.md
