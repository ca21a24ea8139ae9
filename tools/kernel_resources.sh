#!/bin/bash
# VGPRs / scratch / LDS of every kernel in one translation unit:  tools/kernel_resources.sh ndwt_fused3_f64_inv.hip
cd "$(dirname "$0")/../non-decimated_wavelets_amd/csrc" || exit 1
mkdir -p build/asm
o=build/asm/$(basename "$1" .hip).dev.o
hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -simplifycfg-sink-common=false -c --offload-device-only "$1" -o "$o" 2>/dev/null || exit 1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$o" --output="$o.elf" || exit 1
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$o.elf" | grep -E "^\s+\.(name|vgpr_count|agpr_count|private_segment_fixed_size|group_segment_fixed_size|vgpr_spill_count|sgpr_spill_count):" \
  | paste - - - - - - - | sed -E 's/\s+/ /g' | awk '{print}' 
