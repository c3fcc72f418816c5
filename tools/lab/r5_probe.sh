#!/bin/bash
# tools/lab/r5_probe.sh: round 5's first question -- where would an LDS-window decoder stand TODAY?  (i) the round-2 ring kernel against the coop kernel on the three
# payloads; (ii) the coop kernel's Huffman loop alone (decode_only: no copy loads, no stores) on a prefix of the members = at the occupancy an LDS window allows
# (33 % = 3.4 waves per CU, 42 % = 4.3, 50 % = 5.2): decode-only time x (100 / percent) is what a decoder that never waits for memory needs for the whole file.
cd "$(dirname "$0")/../.."
[ -f /tmp/lab50.bam ] || bin/synth_bam write /tmp/lab50.bam 50000000 --seed 1 > /dev/null
[ -f /tmp/labr50.bam ] || bin/synth_bam write /tmp/labr50.bam 50000000 --seed 1 --realistic > /dev/null
[ -f /tmp/labl10.bam ] || bin/synth_bam write /tmp/labl10.bam 10000000 --seed 1 --shape long > /dev/null
show() { python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms'], r['median_ms'], r['members'], r['fnv64_first64MiB'], r['bad_member'])"; }
for f in lab50 labr50 labl10; do
  echo -n "$f coop: "; tools/lab/bin/coop_lab_cur /tmp/$f.bam 8 2>/dev/null | show
  echo -n "$f ring: "; REGTOOLS_AMD_INFLATE=ring tools/lab/bin/coop_lab_cur /tmp/$f.bam 8 2>/dev/null | show
done
for f in lab50 labr50; do
  for v in decode_only noload nostore cur; do
    for pct in 3 25 33 42 50 67 100; do
      echo -n "$f $v $pct%: "; tools/lab/bin/coop_lab_$v /tmp/$f.bam 5 $pct 2>/dev/null | show
    done
  done
done
