#!/bin/bash
# Registers, spills, occupancy, LDS of every kernel (f32 build by default; `tools/resources.sh double` for f64).
# A kernel that suddenly shows 248+ VGPRs and a small code size has stopped inlining its callees: see DESIGN.md §4.
cd "$(dirname "$0")/../raytracing_amd/csrc" || exit 1
REAL=${1:-float}
CONTRACT=on; [ "$REAL" = double ] && CONTRACT=off
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=$CONTRACT -DRTMI_REAL=$REAL \
  -Rpass-analysis=kernel-resource-usage -c -o /dev/null rtmi_kernels.hip 2>&1 |
  grep -E "Function Name|VGPRs:|Spill|Occupancy|LDS Size|ScratchSize" | sed 's/.*remark: //'
