#!/bin/bash
# tools/lab/occ_r4.sh VARIANTS...: the lab binaries on a PREFIX of the bench file's members (3 .. 100 %): how the launch's time follows the number of members in flight
cd "$(dirname "$0")/../.."
[ -f /tmp/lab50.bam ] || bin/synth_bam write /tmp/lab50.bam 50000000 --seed 1 > /dev/null
for v in "$@"; do
  for pct in 3 12 25 33 50 67 75 100; do
    echo -n "$v lab50 $pct%: "; tools/lab/bin/coop_lab_$v /tmp/lab50.bam 6 $pct 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms'], r['median_ms'], r['members'])"
  done
done
