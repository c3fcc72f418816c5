#!/bin/bash
# tools/lab/step_ab.sh: the bench step (host bytes -> table, and device-resident) under environment switches, interleaved
cd "$(dirname "$0")/../.."
for r in 1 2; do
  for v in "A=1" "REGTOOLS_AMD_GATE_SIDE=0" "REGTOOLS_AMD_LITE_WALK=0" "REGTOOLS_AMD_PREAGG=0"; do
    echo -n "$v: "
    env $v python bench.py --steps 8 --warmup 2 --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['ms_per_step'],2), round(d['ms_per_step_device_resident'],2), d['stage_ms'])"
  done
done
