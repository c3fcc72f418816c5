#!/bin/bash
# tools/lab/r5_pieces_sizes.sh: the bench step, the arena (a context's FIRST one, no trials) made of pieces of several sizes, fresh processes, interleaved
cd "$(dirname "$0")/../.."
one() {
  echo -n "$1 $2: "
  env $2 python bench.py --steps ${STEPS:-10} --warmup 3 --no-extras --no-cpu-baseline --no-live-traffic $3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('step', round(d['ms_per_step'],2), 'resident', round(d['ms_per_step_device_resident'],2), 'launch', round(r['kernel_ms'],2), 'trials', r.get('arena_placement_trials_ms'), d['stage_ms'])"
}
for r in $(seq ${ROUNDS:-2}); do
  for v in ${SIZES:-32 64 128 256 512 1024 2048 4096}; do one default "REGTOOLS_AMD_ARENA=${TRIALS:-0},$v" ""; done
done
