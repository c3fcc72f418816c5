#!/bin/bash
# GPU-box script: round 5's evidence -- the GPU suite, the default bench line, rocprofv3 --kernel-trace --stats of each of the three extraction
# workloads on its own (their DEFLATE launches share one kernel symbol now, so one run per workload keeps them apart), the upload / launch timeline.
#   tools/run_r5_evidence.sh [TAG]   ->  gpurun_out/r5/TAG/
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r5/${1:-final}; mkdir -p $O; export TMPDIR=/tmp
if [ -z "$SKIP_SUITE" ]; then timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log | tail -1; fi
cd /tmp
for w in "default:" "realistic:--realistic" "long10M:--shape long --reads 10000000 --steps 2"; do
  key=${w%%:*}; args=${w#*:}
  # (REGTOOLS_AMD_ARENA=0,512: no placement trials in these runs, so that the kernel's row of the stats holds the bench's own launches only)
  REGTOOLS_AMD_ARENA=0,512 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$key -o s -- python $R/bench.py --no-extras --no-cpu-baseline --no-live-traffic $args > $O/bench_${key}_under_rocprofv3.json 2> $O/bench_$key.err
done
cd $R
find $O -name "*kernel_trace.csv" -size +1M -delete
if [ -n "$STATS_ONLY" ]; then exit 0; fi
tools/timeline.sh gpurun_out/r5/${1:-final}/timeline --no-extras --steps 4 --warmup 2 > $O/overlap_timeline.txt 2>&1
find $O -name "*kernel_trace.csv" -size +1M -delete; find $O -name "*memory_copy_trace.csv" -size +1M -delete
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.err
