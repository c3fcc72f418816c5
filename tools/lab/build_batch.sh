#!/bin/bash
# tools/lab/build_batch.sh -- lab binaries of the product's k_inflate (one lane per member) with kCopyBatch changed (bytes copied per trip).
set -e
cd "$(dirname "$0")"
mkdir -p bin
build() {   # name, sed expression on inflate_core.h
  local v=$1; shift
  rm -rf src_$v; mkdir src_$v
  cp ../../regtools_amd/csrc/*.h ../../regtools_amd/csrc/kernels.hip src_$v/
  for e in "$@"; do sed -i "$e" src_$v/inflate_core.h; done
  ( cd src_$v && { cat kernels.hip; echo "#define LAB_VARIANT \"$v\""; cat ../inflate_lab_main.inc; } > lab.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-unused-value -o ../bin/inflate_lab_$v lab.hip ) &
}
build base
for n in 16 32 48 64 96; do build cb$n "s/constexpr uint32_t kCopyBatch = 128;/constexpr uint32_t kCopyBatch = $n;/"; done
wait
ls -la bin | grep "inflate_lab_cb\|inflate_lab_base"
