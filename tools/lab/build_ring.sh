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
# ablations (WRONG output, same decode control flow): where does a trip's time go?
NOFLUSH='s/        if (C.any(want_flush)) C.flush_lines(R, O);/        O.fl = O.hd \& ~127u; (void)want_flush;/'
NOWR1='s/                    R.wr4(ci \* 4u, v0, v1, v2, v3);/                    (void)v0; (void)v1; (void)v2; (void)v3;/'
NOWR2='s/                    R.wr4(ci \* 4u, w0, w1, w2, w3);  /                    (void)w0; (void)w1; (void)w2; (void)w3;/'
NOWR3='s/                    R.wr8(O.hp, lit);/                    (void)lit;/'
NOA='s/                    RGX_NEAR(0) RGX_NEAR(1) RGX_NEAR(2) RGX_NEAR(3) RGX_NEAR(4) RGX_NEAR(5) RGX_NEAR(6) RGX_NEAR(7)/                    RGX_NEAR(0)/'
build noflush "$NOFLUSH"
build nowr "$NOWR1" "$NOWR2" "$NOWR3"
build noflush_nowr "$NOFLUSH" "$NOWR1" "$NOWR2" "$NOWR3"
build decode_only "$NOFLUSH" "$NOWR1" "$NOWR2" "$NOWR3" "$NOA"
wait
ls -la bin | grep inflate_lab
