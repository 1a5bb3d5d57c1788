#!/bin/bash
# VGPRs / scratch of klatt_direct<MODE, CH, WPE> (-DCENSUS_CH=8 -DCENSUS_WPE=4 for the two-workgroups-per-CU build) with only some stages' bodies compiled in: which stage sets the kernel's register count.
#   tools/direct_census.sh [mode] [extra hipcc flags]
cd "$(dirname "$0")/.."
mode=${1:-0}; shift
for st in 1 64 2 4 8 16 32 127; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -mllvm -amdgpu-sched-strategy=max-memory-clause \
     --cuda-device-only -S -DCENSUS_MODE=$mode -DCENSUS_ONLY=$st "$@" tools/direct_census.hip -o /tmp/direct_census_$st.s 2>/dev/null
  echo "stages mask $st: $(grep -E '\.vgpr_count|\.sgpr_count|private_segment_fixed_size|vgpr_spill_count' /tmp/direct_census_$st.s | grep -v '^;' | tr -s ' ' | tr '\n' ' ')"
done
