#!/bin/bash
# A/B of builds of the library on the fused SL kernel (development aid): whole lattice and its first 6 layers
set -e
for v in "$@"; do
  lib=motion_planning_amd/libmplattice$v.so
  for L in 6 64; do
    echo "lib=$lib SL_LAYERS=$L"
    MP_LIB_PATH=$lib SL_LAYERS=$L timeout -k 10 100 python tools/time_sl.py 4096 50
  done
done
