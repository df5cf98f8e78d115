#!/bin/bash
# Registers, scratch and static LDS of every kernel in one HIP source (development aid):
#   tools/kernel_resources.sh motion_planning_amd/csrc/sl_kernels.hip [name filter]
# Compiles the device side only (-Rpass-analysis=kernel-resource-usage) and prints one line per kernel.
src=${1:?source file}
filt=${2:-.}
hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off --cuda-device-only -c -o /dev/null \
    -Rpass-analysis=kernel-resource-usage ${EXTRA_FLAGS:-} "$src" 2>&1 | grep "remark:" |
sed -e 's/ \[-Rpass-analysis=kernel-resource-usage\]//' -e 's/^.*remark: *//' |
awk -F': ' '/^Function Name/ {name=$2} /^VGPRs:/ {v=$2} /^AGPRs/ {a=$2} /^TotalSGPRs/ {s=$2} /^ScratchSize/ {sc=$2} /^Occupancy/ {oc=$2} /^LDS Size/ {print name, "vgpr", v, "agpr", a, "sgpr", s, "scratch", sc, "occ", oc, "lds", $2}' |
c++filt | grep -E "$filt"
