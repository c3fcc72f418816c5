#!/bin/bash
# GPU-box script (round 6): the realistic payload with files in flight -- chip turn on / off, 2 and 3 in flight on 16 hardware queues, then the timeline
# (needs tools/lab/r6_occ2.patch and the REGTOOLS_AMD_CHIP_TURN switch of commit "lab: REGTOOLS_AMD_CHIP_TURN=0" applied: neither ships)
# of both forms.   tools/lab/r6_realistic_sustained.sh  ->  gpurun_out/r6/realistic_sustained/
cd "$(dirname "$0")/../.."
O=gpurun_out/r6/realistic_sustained; mkdir -p $O
export GPU_MAX_HW_QUEUES=16
for turn in 1 0; do
  echo "REGTOOLS_AMD_CHIP_TURN=$turn GPU_MAX_HW_QUEUES=16" >> $O/ab.txt
  REGTOOLS_AMD_CHIP_TURN=$turn timeout 600 python tools/sustained_ab.py --realistic --files 8 --depths 2,3 >> $O/ab.txt 2>> $O/ab.err
done
for turn in 1 0; do
  WINDOW_MS=260 REGTOOLS_AMD_CHIP_TURN=$turn tools/timeline_sustained.sh $O/tl$turn --realistic --files 6 --depths 2 > $O/timeline_turn$turn.txt 2>&1
  rm -rf $O/tl$turn
done
cat $O/ab.txt
