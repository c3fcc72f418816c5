#!/bin/bash
# tools/lab/build_ring.sh -- lab binaries of k_inflate_ring with one constant of inflate_ring.h changed each (timing experiments).
set -e
cd "$(dirname "$0")"
mkdir -p bin
build() {   # name, sed expression on inflate_ring.h
  local v=$1; shift
  rm -rf src_$v; mkdir src_$v
  cp ../../regtools_amd/csrc/*.h ../../regtools_amd/csrc/kernels.hip src_$v/
  for e in "$@"; do sed -i "$e" src_$v/inflate_ring.h; done
  ( cd src_$v && { cat kernels.hip; echo "#define LAB_VARIANT \"$v\""; cat ../inflate_lab_main.inc; } > lab.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-unused-value -o ../bin/inflate_lab_$v lab.hip ) &
}
build base
build batch64 's/constexpr uint32_t kRingBatch = 128;/constexpr uint32_t kRingBatch = 64; /'
build batch32 's/constexpr uint32_t kRingBatch = 128;/constexpr uint32_t kRingBatch = 32; /'
build flush320 's/constexpr uint32_t kFlushAt = 256;/constexpr uint32_t kFlushAt = 320;/'
wait
ls -la bin | grep inflate_lab
# timing-only experiments (WRONG output: the rings of different waves alias): what would more waves per SIMD buy?
buildk() {   # name, sed expression on kernels.hip
  local v=$1; shift
  rm -rf src_$v; mkdir src_$v
  cp ../../regtools_amd/csrc/*.h ../../regtools_amd/csrc/kernels.hip src_$v/
  for e in "$@"; do sed -i "$e" src_$v/kernels.hip; done
  ( cd src_$v && { cat kernels.hip; echo "#define LAB_VARIANT \"$v\""; cat ../inflate_lab_main.inc; } > lab.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-unused-value -o ../bin/inflate_lab_$v lab.hip ) &
}
buildk occ8 's/constexpr uint32_t kRingLdsBytes = kRingLdsDwords \* 4;/constexpr uint32_t kRingLdsBytes = 20480;/' 's/amdgpu_waves_per_eu(1, 1)/amdgpu_waves_per_eu(2, 2)/'
buildk occ12 's/constexpr uint32_t kRingLdsBytes = kRingLdsDwords \* 4;/constexpr uint32_t kRingLdsBytes = 13568;/' 's/amdgpu_waves_per_eu(1, 1)/amdgpu_waves_per_eu(3, 3)/'
wait
