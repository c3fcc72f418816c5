#!/bin/bash
# tools/lab/build_occ.sh -- the lane kernel at two waves per SIMD (256 VGPRs, no scratch spills) instead of three (168 + 31 spilled), with and without preload.patch
set -e
cd "$(dirname "$0")"
mkdir -p bin
for v in occ8 occ8_preload; do
  rm -rf src_$v; mkdir src_$v
  cp ../../regtools_amd/csrc/*.h ../../regtools_amd/csrc/kernels.hip src_$v/
  sed -i '0,/amdgpu_waves_per_eu(3, 3)/s//amdgpu_waves_per_eu(2, 2)/' src_$v/kernels.hip
  if [ $v = occ8_preload ]; then ( cd src_$v && patch -p0 -s < ../preload.patch ); fi
  ( cd src_$v && { cat kernels.hip; echo "#define LAB_VARIANT \"$v\""; cat ../inflate_lab_main.inc; } > lab.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-unused-value -o ../bin/inflate_lab_$v lab.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-unused-value -S --cuda-device-only -o $v.s lab.hip 2>/dev/null ) &
done
wait
grep -A12 "name:.*k_inflateILb0ELb0" src_occ8/occ8.s | grep "vgpr_count\|spill" | head -3
grep -A12 "name:.*k_inflateILb0ELb0" src_occ8_preload/occ8_preload.s | grep "vgpr_count\|spill" | head -3
