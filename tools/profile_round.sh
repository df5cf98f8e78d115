#!/bin/bash
# Kernel-trace statistics plus the counter passes for one build, on the GPU box.
# Usage: tools/profile_round.sh <tag>   (writes gpurun_out/<tag>_*)
# The lattice runs switch off bench.py's extra legs (whole-replan leg, host-to-device variant,
# CPU oracle, the two-stream schedule): they launch the same kernels at other sizes and would mix into the per-kernel
# averages; the counting instance and the two-kernel SL path have kernel names of their own.
set -u
tag=${1:-prof}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- \
    python3 bench.py --steps 20 --warmup 5 --reps 3 --pipeline-steps 0 --h2d-reps 0 --cpu-sample 0 --single-plan 0 --streams 1 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_stats.log || { tail -5 gpurun_out/${tag}_stats.log; exit 1; }
cp $(find gpurun_out/${tag}_stats -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_kernel_stats.csv &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_pipe_stats -- \
    python3 bench.py --workload pipeline --steps 10 --warmup 2 --reps 3 --cpu-sample 0 --single-plan 0 --streams 1 > gpurun_out/${tag}_pipeline_bench.json 2> gpurun_out/${tag}_pipe_stats.log || { tail -5 gpurun_out/${tag}_pipe_stats.log; exit 1; }
cp $(find gpurun_out/${tag}_pipe_stats -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_pipeline_kernel_stats.csv &&
bash tools/pmc_passes.sh gpurun_out/${tag}_pmc &&
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc > gpurun_out/${tag}_pmc_summary.txt &&
bash tools/pmc_pipeline.sh gpurun_out/${tag}_pmc_pipe &&
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_pipe > gpurun_out/${tag}_pipeline_pmc_summary.txt
