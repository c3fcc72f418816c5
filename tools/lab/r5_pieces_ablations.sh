#!/bin/bash
# tools/lab/r5_pieces_ablations.sh: what is left of the DEFLATE launch's memory cost on an arena of 512 MiB pieces?  The ablated kernels of r5_probe.sh (decode_only: no copy loads,
# no stores; noload; nostore) on a hipMalloc arena and on a piecewise one (-DLAB_PIECES=512), whole file and prefixes of the member list, two processes each, interleaved
cd "$(dirname "$0")/../.."
[ -f /tmp/lab50.bam ] || bin/synth_bam write /tmp/lab50.bam 50000000 --seed 1 > /dev/null
[ -f /tmp/labr50.bam ] || bin/synth_bam write /tmp/labr50.bam 50000000 --seed 1 --realistic > /dev/null
show() { python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms'], r['median_ms'])"; }
for rep in 1 2; do
  for v in decode_only noload nostore cur; do
    for pct in 50 100; do
      echo -n "lab50 $v $pct% hipMalloc: "; tools/lab/bin/coop_lab_$v /tmp/lab50.bam 5 $pct 2>/dev/null | show
      echo -n "lab50 $v $pct% pieces:    "; tools/lab/bin/coop_lab_${v}_p /tmp/lab50.bam 5 $pct 2>/dev/null | show
    done
  done
done
for v in decode_only cur; do
  echo -n "labr50 $v hipMalloc: "; tools/lab/bin/coop_lab_$v /tmp/labr50.bam 5 2>/dev/null | show
  echo -n "labr50 $v pieces:    "; tools/lab/bin/coop_lab_${v}_p /tmp/labr50.bam 5 2>/dev/null | show
done
