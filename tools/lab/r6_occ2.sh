#!/bin/bash
# GPU-box script (round 6): k_inflate_coop at two waves per SIMD (REGTOOLS_AMD_INFLATE_OCC=2: eight waves per CU, registers and LDS left on every CU
# (needs tools/lab/r6_occ2.patch and the REGTOOLS_AMD_CHIP_TURN switch of commit "lab: REGTOOLS_AMD_CHIP_TURN=0" applied: neither ships)
# for another file's tail) against three, with and without the chip turn, bench payload and realistic.   ->  gpurun_out/r6/occ2/
cd "$(dirname "$0")/../.."
O=gpurun_out/r6/occ2; mkdir -p $O
for occ in 3 2 3 2; do
  echo "REGTOOLS_AMD_INFLATE_OCC=$occ bench.py --no-extras --no-cpu-baseline --no-live-traffic" >> $O/ab.txt
  REGTOOLS_AMD_INFLATE_OCC=$occ timeout 600 python bench.py --no-extras --no-cpu-baseline --no-live-traffic 2>> $O/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'ms_per_step': round(d['ms_per_step'],3), 'resident': round(d['ms_per_step_device_resident'],3), 'sustained': d['sustained']['ms_per_file'] if d.get('sustained') else None, 'kernel_ms': round(d['roofline']['kernel_ms'],3), 'in_step': round(d['roofline']['kernel_ms_in_step'],3), 'stage_ms': d['stage_ms']}))" >> $O/ab.txt
done
for q in 4 16; do for occ in 3 2; do for turn in 1 0; do
  echo "GPU_MAX_HW_QUEUES=$q REGTOOLS_AMD_INFLATE_OCC=$occ REGTOOLS_AMD_CHIP_TURN=$turn" >> $O/ab.txt
  GPU_MAX_HW_QUEUES=$q REGTOOLS_AMD_INFLATE_OCC=$occ REGTOOLS_AMD_CHIP_TURN=$turn timeout 300 python tools/sustained_ab.py --files 16 --depths $([ $q = 16 ] && echo 2,3 || echo 2) >> $O/ab.txt 2>> $O/ab.err
done; done; done
for q in 4 16; do
  echo "realistic GPU_MAX_HW_QUEUES=$q" >> $O/ab.txt
  GPU_MAX_HW_QUEUES=$q timeout 600 python tools/sustained_ab.py --realistic --files 8 --depths 2 >> $O/ab.txt 2>> $O/ab.err
done
echo "realistic GPU_MAX_HW_QUEUES=16 REGTOOLS_AMD_INFLATE_OCC=2 REGTOOLS_AMD_CHIP_TURN=0" >> $O/ab.txt
GPU_MAX_HW_QUEUES=16 REGTOOLS_AMD_INFLATE_OCC=2 REGTOOLS_AMD_CHIP_TURN=0 timeout 600 python tools/sustained_ab.py --realistic --files 8 --depths 2,3 >> $O/ab.txt 2>> $O/ab.err
cat $O/ab.txt; tail -5 $O/ab.err
