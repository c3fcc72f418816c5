#!/bin/bash
# GPU box: the arena-trials subprocess (see r6_arena_trials_repro.sh) with another process holding all but X GiB of the device -- does a failed allocation end the call
# with an error (expected) or with a signal?   -> gpurun_out/r6s5/arena_pressure/
cd "$(dirname "$0")/../.."
O=gpurun_out/r6s5/arena_pressure; mkdir -p $O
cat > /tmp/hog.py <<'P'
import sys, time, torch
left = float(sys.argv[1])
free, total = torch.cuda.mem_get_info()
n = int(free - left * (1 << 30))
bufs = []
while n > 0:
    k = min(n, 16 << 30); bufs.append(torch.empty(k, dtype=torch.uint8, device='cuda')); n -= k
free2, _ = torch.cuda.mem_get_info()
print('hog holds, free now %.2f GiB' % (free2 / (1 << 30)), flush=True)
time.sleep(float(sys.argv[2]))
P
cat > /tmp/arena_code.py <<'P'
import sys, json, regtools_amd
from regtools_amd import synth
bam, bai, st = synth.generate(10_000_000, shape='short', seed=21)
try:
    ctx = regtools_amd.Context(0); beds, trials = [], []
    for k in range(3):
        je = regtools_amd.JunctionsExtractor(strandness=0, ctx=ctx)
        je.identify_junctions_from_BAM(bam_bytes=bam, bai_bytes=bai)
        assert je.stats['n_records'] == st['n_reads']
        beds.append(je.bed12()); trials.append(ctx.arena_trials())
        sys.stderr.write('CALL %d done\n' % k)
    ctx.close()
    sys.stderr.write('TRIALS ' + json.dumps(trials) + '\n')
except regtools_amd.RegtoolsError as e:
    sys.stderr.write('REGTOOLS ERROR %r\n' % (str(e)[:200],)); sys.exit(7)
P
for left in 16 10 7 5 4 3 2 1; do
  python /tmp/hog.py $left 100 > $O/hog_$left.txt 2>&1 &
  HOG=$!
  for t in $(seq 1 60); do grep -q "hog holds" $O/hog_$left.txt 2>/dev/null && break; sleep 1; done
  for knob in 5 none; do
    if [ $knob = 5 ]; then export REGTOOLS_AMD_ARENA=5; else unset REGTOOLS_AMD_ARENA; fi
    PYTHONPATH=$PWD timeout 120 python -X faulthandler /tmp/arena_code.py > /dev/null 2> $O/left_${left}_$knob.err
    echo "left $left GiB knob $knob rc $? : $(grep -v amdgpu.ids $O/left_${left}_$knob.err | tail -n 2 | tr '\n' ' ' | cut -c1-300)"
  done
  kill $HOG; wait $HOG 2>/dev/null
done
