#!/bin/bash
# Developer helper: registers / scratch / LDS of every kernel in the built library.
set -e
cd "$(dirname "$0")/.."
tmp=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 \
  --input=<(objcopy -O binary --only-section=.hip_fatbin fhe_string_amd/libfhe_string_amd.so /dev/stdout) \
  --output=$tmp/dev.co --unbundle 2>/dev/null || {
  objcopy -O binary --only-section=.hip_fatbin fhe_string_amd/libfhe_string_amd.so $tmp/fat.bin
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fat.bin --output=$tmp/dev.co --unbundle
}
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/dev.co | awk '
  /\.name:/ {name=$2}
  /\.vgpr_count:/ {v=$2}
  /\.agpr_count:/ {a=$2}
  /\.sgpr_count:/ {s=$2}
  /\.private_segment_fixed_size:/ {p=$2}
  /\.group_segment_fixed_size:/ {g=$2}
  /\.vgpr_spill_count:/ {printf "%-40s vgpr=%s agpr=%s sgpr=%s scratch=%s spill=%s\n", substr(name,1,60), v, a, s, p, $2}'
rm -rf $tmp
