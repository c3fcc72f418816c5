#!/bin/bash
# tools/lab/r5_vmm_chunks.sh: the arena mapped from SUBSETS of forty 1 GiB chunks, a per-chunk cost fitted and tested (inflate_lab_main.inc -DLAB_VMM_CHUNKS), the bench file's whole-range DEFLATE launch into each
cd "$(dirname "$0")/../.."
O=gpurun_out/r5/vmm_chunks; mkdir -p $O
[ -f /tmp/lab50.bam ] || bin/synth_bam write /tmp/lab50.bam 50000000 --seed 1 > /dev/null
for i in $(seq 1 ${N:-2}); do
  timeout 600 tools/lab/bin/coop_lab_vchk /tmp/lab50.bam 2 > $O/p$i.json 2> $O/p$i.err
  echo "process $i"; grep -E "window|random|fit|best|worst|failed" $O/p$i.err; cat $O/p$i.json
done
