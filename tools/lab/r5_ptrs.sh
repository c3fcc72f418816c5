#!/bin/bash
# tools/lab/r5_ptrs.sh: the launch's two speeds are a property of the PROCESS (tools/lab/r5_clocks.sh): where do its buffers lie?
cd "$(dirname "$0")/../.."
[ -f /tmp/lab50.bam ] || bin/synth_bam write /tmp/lab50.bam 50000000 --seed 1 > /dev/null
for i in $(seq 1 ${N:-20}); do
  tools/lab/bin/coop_lab_ptr /tmp/lab50.bam 4 2>/tmp/ptr.err | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms'], r['median_ms'], end=' | ')"; grep ptrs /tmp/ptr.err
done
