#!/bin/bash
# PMC passes of the bench for one library tag (k_snet6 work): tools/exp/pmc_s6.sh <tag|main> -> gpurun_out/pmc_<tag>/pmc.md
set -u
T=$1
R=$PWD
if [ "$T" = main ]; then L=$R/nif_amd/libnif_hip.so; else L=$R/nif_amd/libnif_hip_$T.so; fi
O=$R/gpurun_out/pmc_$T
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "FETCH_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "WRITE_SIZE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  i=$((i+1))
  NIF_LIB=$L timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/p$i -o p$i -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $O/p$i.json 2> $O/p$i.err
  echo "$T pass $i rc=$?"
done
cd $R
python tools/pmc_summary.py $O > $O/pmc.md
