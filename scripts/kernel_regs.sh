#!/bin/bash
# usage: scripts/kernel_regs.sh <object or .so under strawboat_amd/csrc/build> <kernel name pattern>
# prints VGPR / spill / LDS / scratch of the gfx950 kernels whose name matches
set -e
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin "$1" $T/fb
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/fb --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/co | awk -v pat="$2" '
/\.name:/ {name=$2}
/\.group_segment_fixed_size:/ {lds=$2}
/\.private_segment_fixed_size:/ {scr=$2}
/\.sgpr_count:/ {sg=$2}
/\.agpr_count:/ {ag=$2}
/\.vgpr_count:/ {vg=$2}
/\.vgpr_spill_count:/ {sp=$2; if (name ~ pat) printf "%s vgpr=%s agpr=%s spill=%s lds=%s scratch=%s sgpr=%s\n", name, vg, ag, sp, lds, scr, sg}
'
rm -rf $T
