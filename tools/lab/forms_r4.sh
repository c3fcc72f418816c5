#!/bin/bash
# tools/lab/forms_r4.sh VARIANTS...: every lab binary in the forms the selector chooses between, on the payload each is for (A/B/A/B)
cd "$(dirname "$0")/../.."
[ -f /tmp/lab50.bam ] || bin/synth_bam write /tmp/lab50.bam 50000000 --seed 1 > /dev/null
[ -f /tmp/labr50.bam ] || bin/synth_bam write /tmp/labr50.bam 50000000 --seed 1 --realistic > /dev/null
[ -f /tmp/labl10.bam ] || bin/synth_bam write /tmp/labl10.bam 10000000 --seed 1 --shape long > /dev/null
run() { echo -n "$1 $2 [$3]: "; env $3 tools/lab/bin/coop_lab_$2 /tmp/$1.bam 8 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms'], r['median_ms'], r['fnv64_first64MiB'], r['bad_member'])"; }
for r in 1 2; do
for v in "$@"; do
  run lab50 $v "X=1"
  run lab50 $v "REGTOOLS_AMD_INFLATE=lane"
  run labr50 $v "X=1"
  run labr50 $v "REGTOOLS_AMD_INFLATE=lane REGTOOLS_AMD_INFLATE_LITS=4"
  run labr50 $v "REGTOOLS_AMD_INFLATE_TUNE=0"
  run labr50 $v "REGTOOLS_AMD_INFLATE_TUNE=1"
  run labl10 $v "X=1"
  run labl10 $v "REGTOOLS_AMD_INFLATE_TUNE=0 REGTOOLS_AMD_INFLATE_PAIRS=0"
  run labl10 $v "REGTOOLS_AMD_INFLATE_TUNE=0"
  run labl10 $v "REGTOOLS_AMD_INFLATE_TUNE=2 REGTOOLS_AMD_INFLATE_PAIRS=0"
  run labl10 $v "REGTOOLS_AMD_INFLATE=lane"
done
done
