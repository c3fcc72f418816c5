#!/bin/bash
# GPU-box script: the new overlap test, the full gpu suite, then the bench line with a host-side trace of one step.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "overlapped" > gpurun_out/r2/pytest_overlap.txt 2>&1; tail -5 gpurun_out/r2/pytest_overlap.txt
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2/bench.json 2> gpurun_out/r2/bench.err; cat gpurun_out/r2/bench.json; tail -3 gpurun_out/r2/bench.err
REGTOOLS_AMD_TRACE=1 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep "rgx trace" | tail -60 > gpurun_out/r2/trace.txt; head -40 gpurun_out/r2/trace.txt
