#!/bin/bash
# GPU-box script (round 6): the shipped pipeline on 4 / 16 / 32 hardware queues (the chip turn is two wide from 16 queues), three payloads.  -> gpurun_out/r6/queues2/
cd "$(dirname "$0")/../.."
O=gpurun_out/r6/queues2; mkdir -p $O
python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu > $O/pytest_pipeline.log 2>&1; echo "pytest rc=$?" >> $O/ab.txt
for rep in 1 2; do for q in 4 16 32; do
  echo "rep $rep bench payload GPU_MAX_HW_QUEUES=$q" >> $O/ab.txt
  GPU_MAX_HW_QUEUES=$q timeout 300 python tools/sustained_ab.py --files 24 --depths $([ $q = 4 ] && echo 2 || echo 2,3) 2>> $O/ab.err | grep pipeline >> $O/ab.txt
done; done
echo "realistic GPU_MAX_HW_QUEUES=16" >> $O/ab.txt
GPU_MAX_HW_QUEUES=16 timeout 600 python tools/sustained_ab.py --realistic --files 8 --depths 2,3 2>> $O/ab.err >> $O/ab.txt
for q in 4 16; do
  echo "long reads (10 M) GPU_MAX_HW_QUEUES=$q: bench.py's own lines" >> $O/ab.txt
  GPU_MAX_HW_QUEUES=$q timeout 900 python bench.py --no-extras --no-cpu-baseline --no-live-traffic --shape long --reads 10000000 --steps 3 2>> $O/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'ms_per_step': round(d['ms_per_step'],3), 'resident': round(d['ms_per_step_device_resident'],3), 'sustained': d['sustained']}))" >> $O/ab.txt
done
cat $O/ab.txt; tail -3 $O/pytest_pipeline.log
