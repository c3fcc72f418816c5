#!/bin/bash
# tools/lab/build_preload.sh -- the lane kernel with the first batch of a new match loaded one trip early (preload.patch), against the product's.
set -e
cd "$(dirname "$0")"
mkdir -p bin
for v in base preload; do
  rm -rf src_$v; mkdir src_$v
  cp ../../regtools_amd/csrc/*.h ../../regtools_amd/csrc/kernels.hip src_$v/
  if [ $v = preload ]; then ( cd src_$v && patch -p0 -s < ../preload.patch ); fi
  ( cd src_$v && { cat kernels.hip; echo "#define LAB_VARIANT \"$v\""; cat ../inflate_lab_main.inc; } > lab.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-unused-value -save-temps=obj -o ../bin/inflate_lab_$v lab.hip ) &
done
wait
ls -la bin | grep "inflate_lab_preload\|inflate_lab_base"
