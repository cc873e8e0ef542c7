#!/bin/bash
# VGPRs / SGPRs / spills / occupancy of every kernel in the HIP library (compile only, no GPU): tools/kernel_resources.sh [-DX ...]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -std=c++17 \
  -Rpass-analysis=kernel-resource-usage "$@" -c -o /dev/null strelka_amd/csrc/strelka_hip.hip 2>&1 |
  awk '/Function Name:/ {n=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /TotalSGPRs:/ {s=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /Occupancy/ {o=$(NF-1)} /SGPRs Spill/ {ss=$(NF-1)} /VGPRs Spill/ {vs=$(NF-1)} /LDS Size/ {print n, "vgpr", v, "sgpr", s, "scratch", sc, "occ", o, "sgpr_spill", ss, "vgpr_spill", vs, "lds", $(NF-1)}' |
  sed -E 's/_ZN3skh//; s/EvT_.*//' | sort | uniq
