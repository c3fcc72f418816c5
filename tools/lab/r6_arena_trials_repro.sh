#!/bin/bash
# GPU box: the subprocess of tests/test_gpu_parity.py::test_arena_placement_trials_leave_the_results_alone in a loop (it died of SIGSEGV once in the fifth session's
# third suite run), with faulthandler and the library's trace; N runs per knob.
#   tools/lab/r6_arena_trials_repro.sh [N] -> gpurun_out/r6s5/arena_repro/
cd "$(dirname "$0")/../.."
O=gpurun_out/r6s5/arena_repro; mkdir -p $O
N=${1:-20}
cat > /tmp/arena_code.py <<'P'
import sys, json, regtools_amd
from regtools_amd import synth
bam, bai, st = synth.generate(10_000_000, shape='short', seed=21)
ctx = regtools_amd.Context(0); beds, trials = [], []
for k in range(3):
    je = regtools_amd.JunctionsExtractor(strandness=0, ctx=ctx)
    je.identify_junctions_from_BAM(bam_bytes=bam, bai_bytes=bai)
    assert je.stats['n_records'] == st['n_reads']
    beds.append(je.bed12()); trials.append(ctx.arena_trials())
    sys.stderr.write('CALL %d done\n' % k)
ctx.close()
sys.stderr.write('TRIALS ' + json.dumps(trials) + '\n')
P
ulimit -c 0
for i in $(seq 1 $N); do
  for knob in 5 none; do
    if [ $knob = 5 ]; then export REGTOOLS_AMD_ARENA=5; else unset REGTOOLS_AMD_ARENA; fi
    PYTHONPATH=$PWD REGTOOLS_AMD_TRACE=1 timeout 300 python -X faulthandler /tmp/arena_code.py > $O/run_${i}_$knob.out 2> $O/run_${i}_$knob.err
    rc=$?
    echo "run $i knob $knob rc $rc"
    if [ $rc -eq 0 ]; then rm -f $O/run_${i}_$knob.out $O/run_${i}_$knob.err; fi
  done
done
