#!/bin/bash
# tools/lab/r5_arena_map.sh: the DEFLATE launch's time against where the arena lies inside allocations much larger than it (inflate_lab_main.inc -DLAB_ARENA_MAP)
cd "$(dirname "$0")/../.."
O=gpurun_out/r5/arena_map; mkdir -p $O
[ -f /tmp/lab50.bam ] || bin/synth_bam write /tmp/lab50.bam 50000000 --seed 1 > /dev/null
for i in $(seq 1 ${N:-3}); do
  timeout 300 tools/lab/bin/coop_lab_amap /tmp/lab50.bam 2 > $O/p$i.json 2> $O/p$i.err
  echo "process $i"; grep -E "block" $O/p$i.err; cat $O/p$i.json
done
