#!/bin/bash
# GPU-box script: tools/lab/build_preload.sh's two binaries on both payloads, whole launch and a wave running alone (6 % of the members)
cd "$(dirname "$0")/../.."
bin/synth_bam write /tmp/s50.bam 50000000 --threads 64 > /dev/null 2>&1
bin/synth_bam write /tmp/r50.bam 50000000 --threads 64 --realistic > /dev/null 2>&1
for f in /tmp/s50.bam /tmp/r50.bam; do
  echo "== $f"
  for v in ${VARIANTS:-base preload}; do
    REGTOOLS_AMD_INFLATE=lane timeout 60 tools/lab/bin/inflate_lab_$v $f 3 2>&1 | cut -c1-150
    REGTOOLS_AMD_INFLATE=lane timeout 60 tools/lab/bin/inflate_lab_$v $f 2 6 2>&1 | cut -c1-150
  done
done
