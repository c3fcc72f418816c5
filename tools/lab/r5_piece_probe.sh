#!/bin/bash
# tools/lab/r5_piece_probe.sh: does a short probe of every 512 MiB piece of a pool tell which pieces make a fast arena?  (inflate_lab_main.inc -DLAB_PIECE_PROBE=64)
cd "$(dirname "$0")/../.."
O=gpurun_out/r5/piece_probe; mkdir -p $O
[ -f /tmp/lab50.bam ] || bin/synth_bam write /tmp/lab50.bam 50000000 --seed 1 > /dev/null
for i in $(seq 1 ${N:-2}); do
  timeout 600 tools/lab/bin/coop_lab_prob /tmp/lab50.bam 2 > $O/p$i.json 2> $O/p$i.err
  echo "process $i"; grep -E "probe|arena from|failed|fault" $O/p$i.err; cat $O/p$i.json
done
