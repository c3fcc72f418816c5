#!/bin/bash
# tools/lab/r5_identify_ab.sh: config 4 `identify` (bench.py's extra) with the arena as pieces / as one hipMalloc block, fresh processes, interleaved
cd "$(dirname "$0")/../.."
for r in $(seq ${ROUNDS:-3}); do
  for v in "REGTOOLS_AMD_ARENA=5,512" "REGTOOLS_AMD_ARENA=5,0"; do
    echo -n "$v: "
    env $v BENCH_EXTRAS=identify python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); i=d['identify_config4']
print(round(i['seconds'],4), i['all_seconds'], i['stage_ms'])"
  done
done
