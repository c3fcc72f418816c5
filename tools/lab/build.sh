#!/bin/bash
# Builds ablated copies of the inflate kernel for timing experiments:  tools/lab/build.sh  ->  tools/lab/bin/inflate_lab_<variant>
# Each variant is the product source with a few lines patched by tools/lab/variants.py (wrong output, same control flow).
set -e
cd "$(dirname "$0")"
mkdir -p bin
for v in $(python3 variants.py --list); do
  rm -rf src_$v; mkdir src_$v
  cp ../../regtools_amd/csrc/*.h ../../regtools_amd/csrc/kernels.hip src_$v/
  python3 variants.py $v src_$v
  ( cd src_$v && { cat kernels.hip; echo "#define LAB_VARIANT \"$v\""; cat ../inflate_lab_main.inc; } > lab.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-unused-value -o ../bin/inflate_lab_$v lab.hip ) &
done
wait
ls -la bin
