#!/bin/bash
# tools/lab/r5_vmm.sh: can the arena's placement be MADE instead of drawn?  hipMemAddressReserve at a chosen alignment + hipMemCreate in chunks of a chosen size
# (inflate_lab_main.inc -DLAB_VMM), the whole-range DEFLATE launch of the bench file into each.
cd "$(dirname "$0")/../.."
O=gpurun_out/r5/vmm; mkdir -p $O
[ -f /tmp/lab50.bam ] || bin/synth_bam write /tmp/lab50.bam 50000000 --seed 1 > /dev/null
for i in $(seq 1 ${N:-2}); do
  timeout 300 tools/lab/bin/coop_lab_vmm /tmp/lab50.bam 2 > $O/p$i.json 2> $O/p$i.err
  echo "process $i"; grep -E "vmm|hipMalloc|failed" $O/p$i.err; cat $O/p$i.json
done
