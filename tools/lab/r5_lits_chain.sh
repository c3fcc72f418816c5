#!/bin/bash
# tools/lab/r5_lits_chain.sh: the four-literal trip rule (reverted from the product) as a LAB build whose mode word comes from LAB_MODE: does a member's own
# chain -- one wave alone, 3 % of the members -- get shorter with a fifth fewer trips?  (The timed step of the random-bases payload ends a chain behind its upload.)
cd "$(dirname "$0")/../.."
[ -f /tmp/labr50.bam ] || bin/synth_bam write /tmp/labr50.bam 50000000 --seed 1 --realistic > /dev/null
[ -f /tmp/lab50.bam ] || bin/synth_bam write /tmp/lab50.bam 50000000 --seed 1 > /dev/null
show() { python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms'], r['median_ms'], r['members'], r['fnv64_first64MiB'], r['bad_member'])"; }
for r in 1 2; do for f in labr50 lab50; do for pct in 3 12 100; do for m in 7 15; do
  echo -n "$f mode $m $pct%: "; LAB_MODE=$m tools/lab/bin/coop_lab_lits4 /tmp/$f.bam 5 $pct 2>/dev/null | show
done; done; done; done
