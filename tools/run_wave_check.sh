#!/bin/bash
# GPU-box script: the three decoder forms on their kernel tests, the whole suite (small inputs now take k_inflate_wave), the smoke run.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "inflate_kernel" 2>&1 | tail -4
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8
