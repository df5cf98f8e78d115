#!/bin/bash
# A/B of builds of the library on the fused SL kernel (development aid): the whole lattice and
# its first layers.  tools/ab_sl.sh "<layer counts>" <suffix> ...: times
# motion_planning_amd/libmplattice<suffix>.so ("" = the build in place) with only the first k
# layers of every scenario, for every k given.
set -e
layers=$1; shift
for v in "$@"; do
  lib=motion_planning_amd/libmplattice$v.so
  for L in $layers; do
    echo "lib=$lib SL_LAYERS=$L"
    MP_LIB_PATH=$lib SL_LAYERS=$L timeout -k 10 100 python tools/time_sl.py 4096 50
  done
done
