#!/bin/bash
# GPU-box script (round 6, last session): files in flight with and without the early tail (REGTOOLS_AMD_EARLY_TAIL=0: one framing / decode pass behind the launch instead of five
# parts under it), 32 hardware queues, 2 and 3 in flight, three interleaved repetitions.   -> gpurun_out/r6/early_tail/
cd "$(dirname "$0")/../.."
O=gpurun_out/r6/early_tail; mkdir -p $O
for rep in 1 2 3; do for et in "6,9,12,14" "0" "8,12" "12"; do
  echo "rep $rep REGTOOLS_AMD_EARLY_TAIL=$et" >> $O/ab.txt
  REGTOOLS_AMD_EARLY_TAIL=$et GPU_MAX_HW_QUEUES=32 timeout 300 python tools/sustained_ab.py --files 24 --depths 2,3 2>> $O/ab.err | grep pipeline >> $O/ab.txt
done; done
cat $O/ab.txt
