#!/bin/bash
# tools/lab/r5_vmm_fill.sh: the arena mapped from pieces at chosen alignments of its virtual address, a filler piece in front (inflate_lab_main.inc -DLAB_VMM_FILL), the bench file's whole-range DEFLATE launch into each
cd "$(dirname "$0")/../.."
O=gpurun_out/r5/vmm_fill; mkdir -p $O
[ -f /tmp/lab50.bam ] || bin/synth_bam write /tmp/lab50.bam 50000000 --seed 1 > /dev/null
for i in $(seq 1 ${N:-2}); do
  timeout 600 tools/lab/bin/coop_lab_vfil /tmp/lab50.bam 2 > $O/p$i.json 2> $O/p$i.err
  echo "process $i"; grep -E "^rep|failed" $O/p$i.err; cat $O/p$i.json
done
