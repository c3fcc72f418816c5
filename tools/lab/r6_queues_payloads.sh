#!/bin/bash
# GPU-box script (round 6): the shipped pipeline on 4 / 16 / 32 hardware queues (launches in turns below 16, at once from 16), three payloads.  -> gpurun_out/r6/queues/
cd "$(dirname "$0")/../.."
O=gpurun_out/r6/queues; mkdir -p $O
python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu > $O/pytest_pipeline.log 2>&1; echo "pytest rc=$?" >> $O/ab.txt
for rep in 1 2; do for q in 4 16 32; do
  echo "rep $rep bench payload GPU_MAX_HW_QUEUES=$q" >> $O/ab.txt
  GPU_MAX_HW_QUEUES=$q timeout 300 python tools/sustained_ab.py --files 24 --depths $([ $q = 4 ] && echo 2 || echo 2,3) 2>> $O/ab.err | grep pipeline >> $O/ab.txt
done; done
for q in 4 16; do
  echo "realistic GPU_MAX_HW_QUEUES=$q" >> $O/ab.txt
  GPU_MAX_HW_QUEUES=$q timeout 600 python tools/sustained_ab.py --realistic --files 8 --depths 2 2>> $O/ab.err >> $O/ab.txt
done
cat $O/ab.txt; tail -3 $O/pytest_pipeline.log
