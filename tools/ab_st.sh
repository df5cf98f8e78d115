#!/bin/bash
# A/B of builds of the library on the benchmark's speed DP (development aid):
# tools/ab_st.sh <suffix> ... times motion_planning_amd/libmplattice<suffix>.so ("" = the build in place).
set -e
for v in "$@"; do
  lib=motion_planning_amd/libmplattice$v.so
  echo "lib=$lib"
  MP_LIB_PATH=$lib ST_AB_ONLY=1 timeout -k 10 100 python tools/time_st_lanes.py 4096
done
