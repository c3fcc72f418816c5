#!/bin/bash
# tools/lab/r5_pieces_ab.sh: the bench step with the arena as one hipMalloc block / as pieces of 512 MiB, with and without the placement trials (REGTOOLS_AMD_ARENA="trials,piece_MiB"),
# fresh processes, interleaved; then the realistic payload and long reads
cd "$(dirname "$0")/../.."
one() { # label, env, extra bench args
  echo -n "$1 $2: "
  env $2 python bench.py --steps ${STEPS:-12} --warmup 3 --no-extras --no-cpu-baseline --no-live-traffic $3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('step', round(d['ms_per_step'],2), 'resident', round(d['ms_per_step_device_resident'],2), 'launch', round(r['kernel_ms'],2), 'in step', r.get('kernel_ms_in_step'), 'trials', r.get('arena_placement_trials_ms'), d['stage_ms'])"
}
for r in $(seq ${ROUNDS:-3}); do
  for v in "REGTOOLS_AMD_ARENA=5,512" "REGTOOLS_AMD_ARENA=5,0" "REGTOOLS_AMD_ARENA=0,512" "REGTOOLS_AMD_ARENA=0,0"; do one default "$v" ""; done
done
for v in "REGTOOLS_AMD_ARENA=5,512" "REGTOOLS_AMD_ARENA=5,0" "REGTOOLS_AMD_ARENA=0,512" "REGTOOLS_AMD_ARENA=0,0"; do one realistic "$v" "--realistic"; done
for v in "REGTOOLS_AMD_ARENA=5,512" "REGTOOLS_AMD_ARENA=5,0" "REGTOOLS_AMD_ARENA=0,512" "REGTOOLS_AMD_ARENA=0,0"; do one long10M "$v" "--shape long --reads 10000000"; done
