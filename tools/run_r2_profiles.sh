#!/bin/bash
# GPU-box script: the round's committed evidence (copied from gpurun_out/r2/prof into profiles/ afterwards).
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r2/prof; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
# 1. per-kernel statistics of the headline workload alone, and of the default command (all workloads)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_noextras -o s -- python $R/bench.py --no-extras > $O/bench_noextras_under_rocprofv3.json 2> $O/bench_noextras.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -o s -- python $R/bench.py > $O/bench_default_under_rocprofv3.json 2> $O/bench_default.err
cd $R
# 2. HBM traffic of the whole-file inflate launch (separate --pmc passes), round 1's kernel and the LDS-window kernel
bash tools/pmc_traffic.sh gpurun_out/r2/prof/traffic_r1 > $O/traffic_r1.txt 2>&1
REGTOOLS_AMD_INFLATE=ring bash tools/pmc_traffic.sh gpurun_out/r2/prof/traffic_ring > $O/traffic_ring.txt 2>&1
# 3. who overlaps whom in the timed region
bash tools/timeline.sh gpurun_out/r2/prof/tl --steps 2 --warmup 2 > $O/overlap_timeline.txt 2>&1
# 4. the plain default line
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default_plain.err
ls -la $O; cat $O/traffic_r1.txt | tail -2; cat $O/traffic_ring.txt | tail -2
