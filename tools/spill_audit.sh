#!/bin/bash
# Lists every kernel of the built library (csrc/build/*.o) whose registers spill to scratch: name, VGPRs, spilled VGPRs.
# A spill reload inside a plane loop waits for every load in flight (vmcnt(0)): complex128 db4 analysis ran 0.97 ms per launch at
# 256^3 on a tile that spilled 16 registers and 0.58 ms on one that does not.   tools/spill_audit.sh [all]
B=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
for o in "$(dirname "$0")"/../non-decimated_wavelets_amd/csrc/build/*.o; do
  b=$(basename "$o" .o)
  $B/llvm-objcopy --dump-section .hip_fatbin=$T/$b.fat "$o" 2>/dev/null || continue
  $B/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/$b.fat --output=$T/$b.elf 2>/dev/null || continue
  $B/llvm-readelf --notes $T/$b.elf 2>/dev/null | grep -E "^\s+\.(name|vgpr_count|vgpr_spill_count):" | paste - - - | sed -E 's/\s+/ /g'
done | awk -v all="$1" 'all == "all" || $NF + 0 > 0' | sed -E 's/ \.name: _ZN4ndwt[0-9]+fused[23]_kernelINS_[0-9]/ /; s/EEEEEvNT_.*(\.vgpr_count)/ \1/'
rm -rf $T
