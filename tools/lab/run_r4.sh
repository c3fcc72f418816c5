#!/bin/bash
# tools/lab/run_r4.sh VARIANTS...: lab binaries on the three payloads (bench 50 M reads, realistic 50 M, long reads 10 M), A/B on one box
cd "$(dirname "$0")/../.."
[ -f /tmp/lab50.bam ] || bin/synth_bam write /tmp/lab50.bam 50000000 --seed 1 > /dev/null
[ -f /tmp/labr50.bam ] || bin/synth_bam write /tmp/labr50.bam 50000000 --seed 1 --realistic > /dev/null
[ -f /tmp/labl10.bam ] || bin/synth_bam write /tmp/labl10.bam 10000000 --seed 1 --shape long > /dev/null
for rep in 1 2; do
for v in "$@"; do
  for f in /tmp/lab50.bam /tmp/labr50.bam /tmp/labl10.bam; do
    echo -n "$v $(basename $f): "; tools/lab/bin/coop_lab_$v $f 3 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms'], r['fnv64_first64MiB'], r['bad_member'])"
  done
done
done
