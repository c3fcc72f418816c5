#!/bin/bash
# tools/lab/r5_identify_lib_ab.sh: config 4 `identify` with two builds of the library (tools/lab/bin/libregtools_amd_before.so / _after.so), fresh processes, interleaved
cd "$(dirname "$0")/../.."
for r in $(seq ${ROUNDS:-4}); do
  for v in before after; do
    cp tools/lab/bin/libregtools_amd_$v.so regtools_amd/libregtools_amd.so
    echo -n "$v: "
    BENCH_EXTRAS=identify python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); i=d['identify_config4']
print(round(i['seconds'],4), i['all_seconds'], i['stage_ms'], i['reference']['outputs_identical_to_gpu'])"
  done
done
cp tools/lab/bin/libregtools_amd_after.so regtools_amd/libregtools_amd.so
