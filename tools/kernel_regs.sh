#!/bin/bash
# Per-kernel register / LDS / occupancy report of one HIP source (hipcc remarks; no GPU needed):
#   tools/kernel_regs.sh upsp_processing_amd/csrc/raycast.hip [filter-regex]
src=$1; filt=${2:-.}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -Iupsp_processing_amd/csrc \
  --offload-arch=gfx950 -c "$src" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/remark: Function Name:/ {name=$(NF-1)}
       /remark:     TotalSGPRs:/ {sg=$(NF-1)}
       /remark:     VGPRs:/ {v=$(NF-1)}
       /remark:     ScratchSize/ {sc=$(NF-1)}
       /remark:     Occupancy/ {oc=$(NF-1)}
       /remark:     VGPRs Spill:/ {sp=$(NF-1)}
       /remark:     LDS Size/ {print name, "vgpr", v, "sgpr", sg, "spill", sp, "scratch", sc, "occ", oc, "lds", $(NF-1)}' |
  c++filt | sed 's/upsp::(anonymous namespace):://; s/(.*) vgpr/ vgpr/' | grep -E "$filt"
