#!/bin/bash
# tools/lab/step_ab.sh [VARIANTS...]: the bench step (host bytes -> table, and device-resident) under environment switches, interleaved
cd "$(dirname "$0")/../.."
if [ $# -eq 0 ]; then set -- "A=1" "REGTOOLS_AMD_EARLY_TAIL=0" "REGTOOLS_AMD_EARLY_TAIL=12" "REGTOOLS_AMD_EARLY_TAIL=10,13" "REGTOOLS_AMD_EARLY_TAIL=6,10,13"; fi
for r in $(seq ${ROUNDS:-2}); do
  for v in "$@"; do
    echo -n "$v: "
    env $v python bench.py --steps ${STEPS:-12} --warmup 3 --no-extras --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['ms_per_step'],2), round(d['ms_per_step_device_resident'],2), d['stage_ms'])"
  done
done
