#!/bin/bash
# GPU-box script (round 6): the pipeline's forms, three interleaved repetitions on one box (24 files per line).   ->  gpurun_out/r6/occ2_rep/
# (needs tools/lab/r6_occ2.patch and the REGTOOLS_AMD_CHIP_TURN switch of commit "lab: REGTOOLS_AMD_CHIP_TURN=0" applied: neither ships)
cd "$(dirname "$0")/../.."
O=gpurun_out/r6/occ2_rep; mkdir -p $O
for rep in 1 2 3; do
  for cfg in "4 3 1" "16 3 1" "16 3 0" "16 2 0" "16 2 1" "8 2 0" "8 3 0"; do
    set -- $cfg
    echo "rep $rep GPU_MAX_HW_QUEUES=$1 REGTOOLS_AMD_INFLATE_OCC=$2 REGTOOLS_AMD_CHIP_TURN=$3" >> $O/ab.txt
    GPU_MAX_HW_QUEUES=$1 REGTOOLS_AMD_INFLATE_OCC=$2 REGTOOLS_AMD_CHIP_TURN=$3 timeout 300 python tools/sustained_ab.py --files 24 --depths 2 2>> $O/ab.err | grep pipeline >> $O/ab.txt
  done
done
cat $O/ab.txt
