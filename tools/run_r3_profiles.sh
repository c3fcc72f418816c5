#!/bin/bash
# GPU-box script: round 3's committed evidence (copied from gpurun_out/r3/prof into profiles/ afterwards).
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r3/prof; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
# 1. per-kernel statistics of the headline workload (the default command without the other workloads), and the line it printed
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_noextras -o s -- python $R/bench.py --no-extras > $O/bench_noextras_under_rocprofv3.json 2> $O/bench_noextras.err
cd $R
# 2. HBM-side traffic + instruction counters of the whole-file inflate launch: separate --pmc passes over the kernel alone (tools/inflate_bench.py)
bash tools/pmc_inflate.sh gpurun_out/r3/prof/pmc 1,4 --reads 50000000 > $O/pmc_inflate.txt 2>&1
# 3. who overlaps whom in the timed region
bash tools/timeline.sh gpurun_out/r3/prof/tl --steps 2 --warmup 2 --no-extras > $O/overlap_timeline.txt 2>&1
# 4. the plain default line (all workloads)
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
# 5. the lab's ablations of k_inflate_coop on this box
tools/lab/run_coop.sh base noload loadsonly decode_only nostore > $O/inflate_coop_lab.txt 2>&1
ls -la $O; tail -3 $O/pmc_inflate.txt
