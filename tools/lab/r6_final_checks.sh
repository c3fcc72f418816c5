#!/bin/bash
# GPU-box script (round 6, last session): soak of the pipeline on 32 / 16 / 4 hardware queues, the sustained pass's timeline as bench.py runs it, the GPU suite.
cd "$(dirname "$0")/../.."
O=gpurun_out/r6/final_checks; mkdir -p $O
GPU_MAX_HW_QUEUES=32 timeout 900 python tools/soak/pipeline_soak.py 240 2,3 > $O/soak.txt 2> $O/soak.err
GPU_MAX_HW_QUEUES=16 timeout 900 python tools/soak/pipeline_soak.py 240 2,3 >> $O/soak.txt 2>> $O/soak.err
GPU_MAX_HW_QUEUES=4 timeout 900 python tools/soak/pipeline_soak.py 240 2 >> $O/soak.txt 2>> $O/soak.err
cat $O/soak.txt; tail -3 $O/soak.err
tools/timeline_sustained.sh $O/tl --depths 2 --files 8 > $O/sustained_timeline.txt 2>&1; rm -rf $O/tl
python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc=$?"; tail -2 $O/gputests.log
