#!/bin/bash
# GPU-box script: the product's lane kernel with other copy batch sizes on the two bench payloads (REGTOOLS_AMD_INFLATE=lane: the lab's launch is the lane form)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r2
O=gpurun_out/r2/batch_lab.txt
: > $O
bin/synth_bam write /tmp/s50.bam 50000000 --threads 64 >> $O 2>&1
bin/synth_bam write /tmp/r50.bam 50000000 --threads 64 --realistic >> $O 2>&1
for f in /tmp/s50.bam /tmp/r50.bam; do
  echo "== $f" >> $O
  for v in base cb16 cb32 cb48 cb64 cb96; do
    REGTOOLS_AMD_INFLATE=lane timeout 60 tools/lab/bin/inflate_lab_$v $f 3 >> $O 2>&1
  done
done
cat $O
