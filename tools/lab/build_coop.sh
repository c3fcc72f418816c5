#!/bin/bash
# Ablated copies of round 3's k_inflate_coop for timing experiments:  tools/lab/build_coop.sh [variants...]  ->  tools/lab/bin/coop_lab_<variant>
set -e
cd "$(dirname "$0")"
mkdir -p bin
VS="$@"; [ -z "$VS" ] && VS=$(python3 variants_coop.py --list)
for v in $VS; do
  rm -rf src_$v; mkdir src_$v
  cp ../../regtools_amd/csrc/*.h ../../regtools_amd/csrc/kernels.hip src_$v/
  python3 variants_coop.py $v src_$v
  ( cd src_$v && { cat kernels.hip; echo "#define LAB_VARIANT \"$v\""; cat ../inflate_lab_main.inc; } > lab.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-unused-value -o ../bin/coop_lab_$v lab.hip ) &
done
wait
ls bin | grep coop_lab
