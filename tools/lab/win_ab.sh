#!/bin/bash
# tools/lab/win_ab.sh [ROUNDS]: k_inflate_coop with the 16-byte window (unaligned loads) against the 32-byte window of aligned loads
# (REGTOOLS_AMD_INFLATE_TUNE bit 1 / bit 2; bit 0 = lanes sorted by compressed length), interleaved, three payloads.
cd "$(dirname "$0")/../.."
N=${1:-2}
run() {   # name reads extra mode tunes...
  local name=$1 reads=$2 extra=$3 mode=$4; shift 4
  for r in $(seq $N); do
    for t in "$@"; do
      echo -n "$name mode $mode tune $t: "
      REGTOOLS_AMD_INFLATE_PAIRS=$mode REGTOOLS_AMD_INFLATE_TUNE=$t python tools/inflate_bench.py --reads $reads $extra --forms 4 --reps 5 --bam /tmp/$name.bam 2>/dev/null | grep ms_min | head -1
    done
  done
}
run lab50 50000000 "" 7 3 5
run labr50 50000000 "--realistic" 7 3 5
run labl10 10000000 "--shape=long" 4 2 4
