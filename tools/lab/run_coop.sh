#!/bin/bash
# tools/lab/run_coop.sh [variants...]: the coop ablations on the bench file (synthetic, 50 M reads) and on 10 M reads of the realistic payload
cd "$(dirname "$0")/../.."
[ -f /tmp/lab50.bam ] || bin/synth_bam write /tmp/lab50.bam 50000000 --seed 1 > /dev/null
[ -f /tmp/labr10.bam ] || bin/synth_bam write /tmp/labr10.bam 10000000 --seed 1 --realistic > /dev/null
VS="$@"; [ -z "$VS" ] && VS=$(python3 tools/lab/variants_coop.py --list)
for v in $VS; do
  for f in /tmp/lab50.bam /tmp/labr10.bam; do
    echo -n "$v $(basename $f) full: "; tools/lab/bin/coop_lab_$v $f 3 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms'], r['fnv64_first64MiB'], r['bad_member'])"
  done
  echo -n "$v lab50 alone(3%): "; tools/lab/bin/coop_lab_$v /tmp/lab50.bam 3 3 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms'])"
done
