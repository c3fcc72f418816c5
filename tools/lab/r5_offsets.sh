#!/bin/bash
# tools/lab/r5_offsets.sh: the launch's speed is a property of the arena (tools/lab/inflate_lab_main.inc -DLAB_MULTI_ARENA): of its PAGES, or of its address inside them?
# Eight arenas per process, then the fastest and the slowest slid inside their own allocations (-DLAB_ARENA_OFFSETS).
cd "$(dirname "$0")/../.."
O=gpurun_out/r5/offsets; mkdir -p $O
[ -f /tmp/lab50.bam ] || bin/synth_bam write /tmp/lab50.bam 50000000 --seed 1 > /dev/null
for i in $(seq 1 ${N:-3}); do
  timeout 300 tools/lab/bin/coop_lab_moff /tmp/lab50.bam 2 > $O/p$i.json 2> $O/p$i.err
  echo "process $i"; grep -E "^pass|^offset|^input" $O/p$i.err
done
