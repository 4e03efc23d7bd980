#!/bin/bash
# Register / LDS / scratch figures of the kernels of one source file, from the code object notes (no GPU needed):
#   bash scripts/kernel_resources.sh kernels_tiles inner_iterations kernels_bcr ...      (names under openimucameracalibrator_amd/csrc, without .hip)
R=$(cd $(dirname $0)/.. && pwd); T=$(mktemp -d)
for f in "$@"; do
  extra=""; [ "$f" = kernels_tiles ] && extra="-mllvm -amdgpu-mfma-vgpr-form"    # (as the Makefile builds it)
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics $extra --cuda-device-only -c $R/openimucameracalibrator_amd/csrc/$f.hip -o $T/$f.co 2>/dev/null || { echo "$f: compile failed"; continue; }
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/$f.co --targets=hip-amdgcn-amd-amdhsa--gfx950 --output=$T/$f.elf
  echo "# $f.hip: kernel, vgpr_count (arch + acc), agpr_count, vgpr_spill_count, sgpr_spill_count, scratch bytes (private_segment_fixed_size), static LDS bytes (group_segment_fixed_size)"
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/$f.elf | grep -E "\.name:|\.vgpr_count|\.agpr_count|private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size" | paste - - - - - - - \
    | sed -E 's/.*agpr_count: *([0-9]+).*group_segment_fixed_size: *([0-9]+).*\.name: *([^ \t]+).*private_segment_fixed_size: *([0-9]+).*sgpr_spill_count: *([0-9]+).*\.vgpr_count: *([0-9]+).*vgpr_spill_count: *([0-9]+).*/\3,\6,\1,\7,\5,\4,\2/' | while IFS=, read n a b c d e g; do echo "$(echo $n | c++filt | cut -c1-110),$a,$b,$c,$d,$e,$g"; done
done
rm -rf $T
