#!/bin/bash
# rocprofv3 counter passes over the whole-replan workload (bench.py --workload pipeline).
# Usage: tools/pmc_pipeline.sh <out_dir>
set -u
out=${1:-gpurun_out/pmc_pipe}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py --workload pipeline --steps 3 --warmup 1 --reps 1 --cpu-sample 0 --single-plan 0 --streams 1 > "$out.$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$out.$name.log"; exit 1; }
  echo "pass $name ok"
}
mkdir -p "$out"
run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT &&
run sq2 SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
