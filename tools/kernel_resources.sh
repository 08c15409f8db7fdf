#!/bin/bash
# Prints VGPRs / spills / scratch / LDS / occupancy of every kernel in rt_device.hip (compile-only, no GPU needed).
cd "$(dirname "$0")/../ray_tracing_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize --offload-arch=gfx950 "$@" -c rt_device.hip -o /tmp/rt_res.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|VGPR Spill|ScratchSize|Occupancy|LDS Size" \
  | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - - | sed -E 's/Function Name: _ZN3rta//; s/ENS_12RenderParamsE//'
