#!/bin/bash
# GPU-box script: A/B of round 1's k_inflate (REGTOOLS_AMD_INFLATE_R1=1) against k_inflate_ring variants on the bench files.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/ring
O=gpurun_out/ring/lab.txt
: > $O
bin/synth_bam write /tmp/s50.bam 50000000 --threads 64 >> $O 2>&1
bin/synth_bam write /tmp/r50.bam 50000000 --threads 64 --realistic >> $O 2>&1
for f in /tmp/s50.bam /tmp/r50.bam; do
  echo "== $f" >> $O
  timeout 60 tools/lab/bin/inflate_lab_base $f 3 >> $O 2>&1
  for v in ${VARIANTS:-base batch64 batch32 flush320}; do
    REGTOOLS_AMD_INFLATE=ring timeout 60 tools/lab/bin/inflate_lab_$v $f 3 >> $O 2>&1
    REGTOOLS_AMD_INFLATE=ring timeout 60 tools/lab/bin/inflate_lab_$v $f 2 6 >> $O 2>&1
  done
done
cat $O
