#!/bin/bash
# tools/lab/runs_ab.sh [ROUNDS]: k_inflate_coop's mode word (inflate_coop.h: bit 0 a literal + the next symbol per trip, bit 1 a literal pair + a match
# per trip with exact bit counts, bit 2 runs written from registers), interleaved, three payloads, through the stage entry point
# (tools/inflate_bench.py checks a sample of members against zlib).
cd "$(dirname "$0")/../.."
N=${1:-2}
mkdir -p gpurun_out
run() {   # name reads extra modes...
  local name=$1 reads=$2 extra=$3; shift 3
  for r in $(seq $N); do
    for m in "$@"; do
      echo -n "$name mode $m: "
      REGTOOLS_AMD_INFLATE_PAIRS=$m python tools/inflate_bench.py --reads $reads $extra --forms 4 --reps 5 --bam /tmp/$name.bam 2>/dev/null | grep ms_min | head -1
    done
  done
}
run lab50 50000000 "" ${MODES_BENCH:-1 3 7}
run labr50 50000000 "--realistic" ${MODES_REAL:-1 3}
run labl10 10000000 "--shape=long" ${MODES_LONG:-0 1 4 5 7}
