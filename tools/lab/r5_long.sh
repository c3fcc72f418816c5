#!/bin/bash
# tools/lab/r5_long.sh: what a sector stage could be worth on a MULTI-ROUND payload (long reads: 15.5 k waves) at the occupancy it would run at: the product kernel, with
# three of four stage stores dropped (wrong output: the bound), and both with the whole symbol list in LDS = 7 waves per CU (what 128 B of stage per lane leave).
cd "$(dirname "$0")/../.."
[ -f /tmp/labl10.bam ] || bin/synth_bam write /tmp/labl10.bam 10000000 --seed 1 --shape long > /dev/null
show() { python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms'], r['median_ms'], r['members'], r['fnv64_first64MiB'], r['bad_member'])"; }
for r in 1 2; do
  for v in cur drop h288 h288drop; do
    echo -n "labl10 $v: "; REGTOOLS_AMD_INFLATE=coop tools/lab/bin/coop_lab_$v /tmp/labl10.bam 6 2>/dev/null | show
  done
done
