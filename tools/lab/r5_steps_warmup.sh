#!/bin/bash
# tools/lab/r5_steps_warmup.sh: bench.py's timed step against its --steps / --warmup (is one warm-up step the steady state?), fresh processes, interleaved
cd "$(dirname "$0")/../.."
for r in 1 2 3; do
  for sw in "3 1" "3 2" "3 3" "10 3" "10 1"; do
    set -- $sw
    echo -n "steps $1 warmup $2: "
    python bench.py --steps $1 --warmup $2 --no-extras --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), round(d['ms_per_step_device_resident'],2))"
  done
done
