#!/bin/bash
# tools/lab/ab_r4.sh ROUNDS VARIANTS...: interleaved A/B/A/B of lab binaries (12 launches each per round; min and median), three payloads
cd "$(dirname "$0")/../.."
N=$1; shift
[ -f /tmp/lab50.bam ] || bin/synth_bam write /tmp/lab50.bam 50000000 --seed 1 > /dev/null
[ -f /tmp/labr50.bam ] || bin/synth_bam write /tmp/labr50.bam 50000000 --seed 1 --realistic > /dev/null
[ -f /tmp/labl10.bam ] || bin/synth_bam write /tmp/labl10.bam 10000000 --seed 1 --shape long > /dev/null
for f in lab50 labr50 labl10; do
  for r in $(seq $N); do
    for v in "$@"; do
      echo -n "$f $v: "; tools/lab/bin/coop_lab_$v /tmp/$f.bam 12 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms'], r['median_ms'], r['fnv64_first64MiB'], r['bad_member'])"
    done
  done
done
