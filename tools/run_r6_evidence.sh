#!/bin/bash
# GPU-box script: round 6's evidence -- rocprofv3 --kernel-trace --stats of each of the three extraction workloads on its own (their DEFLATE launches
# share one kernel symbol, so one run per workload keeps them apart), each workload's bench line with the PMC traffic measured in the run, the
# default bench line (all extras), the sustained pass's timeline.
#   tools/run_r6_evidence.sh [TAG]   ->  gpurun_out/r6/TAG/
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r6/${1:-final}; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
for w in "default:" "realistic:--realistic" "long10M:--shape long --reads 10000000 --steps 2"; do
  key=${w%%:*}; args=${w#*:}
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$key -o s -- python $R/bench.py --no-extras --no-cpu-baseline --no-live-traffic --no-sustained $args > $O/bench_${key}_under_rocprofv3.json 2> $O/bench_$key.err
done
cd $R
find $O -name "*kernel_trace.csv" -size +1M -delete
for w in "realistic:--realistic" "long10M:--shape long --reads 10000000 --steps 2"; do
  key=${w%%:*}; args=${w#*:}
  timeout 1200 python bench.py --no-extras --no-cpu-baseline $args > $O/bench_${key}.json 2> $O/bench_${key}_plain.err
done
tools/timeline_sustained.sh gpurun_out/r6/${1:-final}/timeline --depths 2 --files 8 > $O/sustained_timeline.txt 2>&1
rm -rf $O/timeline
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.err
