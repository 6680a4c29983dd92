#!/bin/bash
# measurement builds of k_snet6 (tools/build_variant.py) -- run on the CPU box, then tools/exp/ab_libs.sh on the GPU
set -e
python tools/build_variant.py s6noring "-DNIF_S6_RING=0" k_snet6.hip
python tools/build_variant.py s6nocons "-DNIF_S6_NOCONS" k_snet6.hip
python tools/build_variant.py s6nodep "-DNIF_S6_NOCONS -DNIF_S6_NODEP" k_snet6.hip
python tools/build_variant.py s6nostash "-DNIF_ABL_NOSTORE -DNIF_ABL_NOLOAD" k_snet6.hip
python tools/build_variant.py s6bare "-DNIF_S6_NOCONS -DNIF_S6_NODEP -DNIF_ABL_NOSTORE -DNIF_ABL_NOLOAD" k_snet6.hip
