#!/bin/bash
# tools/lab/build_cur.sh NAME [extra hipcc flags]: the CURRENT product kernels + the lab's timing main -> tools/lab/bin/coop_lab_NAME
# (keeps older binaries next to it for A/B runs on one box)
set -e
cd "$(dirname "$0")"
mkdir -p bin
v=$1; shift
rm -rf src_$v; mkdir src_$v
cp ../../regtools_amd/csrc/*.h ../../regtools_amd/csrc/kernels.hip src_$v/
( cd src_$v && { cat kernels.hip; echo "#define LAB_VARIANT \"$v\""; cat ../inflate_lab_main.inc; } > lab.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-unused-value "$@" -o ../bin/coop_lab_$v lab.hip )
ls -la bin/coop_lab_$v
