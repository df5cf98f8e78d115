#!/bin/bash
# rocprofv3 counter passes over the benchmark's speed DP alone (tools/time_st_lanes.py), one
# counter group per run.  Usage: tools/pmc_st.sh <out_dir>; summary on stdout.
set -u
out=${1:-gpurun_out/pmc_st}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export ST_AB_ONLY=1
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 tools/time_st_lanes.py 4096 > "$out.$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$out.$name.log"; exit 1; }
}
mkdir -p "$out"
run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT &&
run sq2 SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE &&
python3 tools/pmc_summary.py "$out"
