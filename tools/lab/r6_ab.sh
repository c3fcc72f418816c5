#!/bin/bash
# tools/lab/r6_ab.sh V1 V2 ...: lab binaries tools/lab/bin/coop_lab_V* interleaved on one box, the three payloads (bench, random bases + binned qualities, long reads).
cd "$(dirname "$0")/../.."
[ -f /tmp/lab50.bam ] || bin/synth_bam write /tmp/lab50.bam 50000000 --seed 1 > /dev/null
[ -f /tmp/labr50.bam ] || bin/synth_bam write /tmp/labr50.bam 50000000 --seed 1 --realistic > /dev/null
[ -f /tmp/labl10.bam ] || bin/synth_bam write /tmp/labl10.bam 10000000 --seed 1 --shape long > /dev/null
show() { python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms'], r['median_ms'], r['members'], r['fnv64_first64MiB'], r['bad_member'])"; }
for r in 1 2 3; do for f in lab50 labr50 labl10; do for v in "$@"; do echo -n "$f $v: "; tools/lab/bin/coop_lab_$v /tmp/$f.bam 8 2>/dev/null | show; done; done; done
