#!/bin/bash
# tools/lab/r5_lits.sh: up to four literals and the match behind them per trip (k_inflate_coop's mode bit 3, round 5).  (i) the kernel before (coop_lab_cur, built from the
# commit before) and after (coop_lab_lits) on the payloads whose mode did not change; (ii) forms 4 (two literals) and 6 (four) of the stage entry point on all three payloads.
cd "$(dirname "$0")/../.."
[ -f /tmp/lab50.bam ] || bin/synth_bam write /tmp/lab50.bam 50000000 --seed 1 > /dev/null
[ -f /tmp/labr50.bam ] || bin/synth_bam write /tmp/labr50.bam 50000000 --seed 1 --realistic > /dev/null
[ -f /tmp/labl10.bam ] || bin/synth_bam write /tmp/labl10.bam 10000000 --seed 1 --shape long > /dev/null
show() { python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms'], r['median_ms'], r['members'], r['fnv64_first64MiB'], r['bad_member'])"; }
for r in 1 2; do for f in lab50 labr50 labl10; do for v in cur lits; do echo -n "$f $v: "; tools/lab/bin/coop_lab_$v /tmp/$f.bam 8 2>/dev/null | show; done; done; done
for r in 1 2; do
python tools/inflate_bench.py --reads 50000000 --forms 4,6 --reps 5 --bam /tmp/lab50.bam 2>/dev/null | cut -c1-260
python tools/inflate_bench.py --reads 50000000 --realistic --forms 4,6 --reps 5 --bam /tmp/labr50.bam 2>/dev/null | cut -c1-260
done
