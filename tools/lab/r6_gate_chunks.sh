#!/bin/bash
# GPU-box script (round 6, last session): the upload's gate chunks (REGTOOLS_AMD_OVERLAP=min_bytes,chunks: 16 by default) -- 16 / 32 / 64, three interleaved repetitions, the
# timed step of the bench payload and of the realistic payload.   -> gpurun_out/r6/gate_chunks/
cd "$(dirname "$0")/../.."
O=gpurun_out/r6/gate_chunks; mkdir -p $O
for rep in 1 2 3; do for ch in 16 32 64; do
  echo "rep $rep chunks $ch" >> $O/ab.txt
  REGTOOLS_AMD_OVERLAP=8388608,$ch timeout 300 python bench.py --no-extras --no-cpu-baseline --no-live-traffic --no-sustained --steps 20 2>> $O/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'ms_per_step': round(d['ms_per_step'],3), 'in_step': round(d['roofline']['kernel_ms_in_step'],3), 'resident': round(d['ms_per_step_device_resident'],3)}))" >> $O/ab.txt
done; done
for ch in 16 64 16 64; do
  echo "realistic chunks $ch" >> $O/ab.txt
  REGTOOLS_AMD_OVERLAP=8388608,$ch timeout 400 python bench.py --realistic --no-extras --no-cpu-baseline --no-live-traffic --no-sustained --steps 6 2>> $O/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'ms_per_step': round(d['ms_per_step'],3), 'in_step': round(d['roofline']['kernel_ms_in_step'],3), 'resident': round(d['ms_per_step_device_resident'],3)}))" >> $O/ab.txt
done
cat $O/ab.txt
