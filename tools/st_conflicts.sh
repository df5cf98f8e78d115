#!/bin/bash
# LDS bank conflicts of the speed DP and whether they cost time (development aid): the kernel's
# time and its conflict counters with the library in place and with a timing-only build
# (motion_planning_amd/libmplattice_stng.so, -DST_EXP_NOGATHER: the acceleration / jerk table
# reads go to one address per cell instead of the data-dependent ones; results are wrong).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/st_conf
mkdir -p $out
for v in "" _stng; do
  lib=motion_planning_amd/libmplattice$v.so
  echo "== $lib"
  MP_LIB_PATH=$lib timeout -k 10 100 python3 tools/time_st.py 2>&1 | grep "bench shape"
  MP_LIB_PATH=$lib timeout -s KILL 100 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $out/pmc$v -- python3 tools/time_st.py > $out/pmc$v.log 2>&1 || { echo "pmc pass failed"; tail -3 $out/pmc$v.log; exit 1; }
  python3 tools/pmc_summary.py $out/pmc$v | grep "st_dp_r_kernel<5, 4>"
done
