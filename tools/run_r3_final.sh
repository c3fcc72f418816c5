#!/bin/bash
# GPU-box script: the round's closing evidence -- the GPU suite, the default bench line, the per-kernel statistics of the headline workload
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r3/final; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_suite.log 2>&1; tail -1 $O/gpu_suite.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_noextras -o s -- python $R/bench.py --no-extras > $O/bench_noextras_under_rocprofv3.json 2> $O/bench_noextras.err
cd $R
find $O -name "*kernel_trace.csv" -size +1M -delete
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 400 $O/bench_default.err
