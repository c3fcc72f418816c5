#!/bin/bash
# tools/lab/build_hot.sh -- the lane kernel with more of the literal/length symbol list in LDS (fewer lookups in the global cold list, fewer waves per CU)
set -e
cd "$(dirname "$0")"
mkdir -p bin
for n in 128 199 224 288; do
  v=hot$n
  rm -rf src_$v; mkdir src_$v
  cp ../../regtools_amd/csrc/*.h ../../regtools_amd/csrc/kernels.hip src_$v/
  sed -i "s/constexpr uint32_t kHotSyms = 160;/constexpr uint32_t kHotSyms = $n;/" src_$v/kernels.hip
  ( cd src_$v && { cat kernels.hip; echo "#define LAB_VARIANT \"$v\""; cat ../inflate_lab_main.inc; } > lab.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-unused-value -o ../bin/inflate_lab_$v lab.hip ) &
done
wait
ls bin | grep inflate_lab_hot
