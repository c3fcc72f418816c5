#!/bin/bash
# tools/lab/r5_probe2.sh: the same prefixes with the WHOLE symbol list in LDS (-DRGX_LAB_HOT_SYMS=288: no cold-symbol load in the middle of a trip; 7 waves per CU):
# how long is a trip that never waits for global memory inside its decode?
cd "$(dirname "$0")/../.."
[ -f /tmp/lab50.bam ] || bin/synth_bam write /tmp/lab50.bam 50000000 --seed 1 > /dev/null
[ -f /tmp/labr50.bam ] || bin/synth_bam write /tmp/labr50.bam 50000000 --seed 1 --realistic > /dev/null
show() { python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms'], r['median_ms'], r['members'], r['fnv64_first64MiB'], r['bad_member'])"; }
for f in lab50 labr50; do
  for v in h288_decode_only h288_cur2 decode_only cur; do
    for pct in 1 3 25 42 50 58 67; do
      echo -n "$f $v $pct%: "; tools/lab/bin/coop_lab_$v /tmp/$f.bam 5 $pct 2>/dev/null | show
    done
  done
done
