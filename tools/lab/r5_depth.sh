#!/bin/bash
# tools/lab/r5_depth.sh: the DEFLATE launch into an arena made when X GiB of device memory are already taken (inflate_lab_main.inc -DLAB_DEPTH)
cd "$(dirname "$0")/../.."
O=gpurun_out/r5/depth; mkdir -p $O
[ -f /tmp/lab50.bam ] || bin/synth_bam write /tmp/lab50.bam 50000000 --seed 1 > /dev/null
for i in 1 2; do timeout 600 tools/lab/bin/coop_lab_depth /tmp/lab50.bam 2 > $O/p$i.json 2> $O/p$i.err; echo "process $i"; grep -E "^pass|failed|fault" $O/p$i.err; done
