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
