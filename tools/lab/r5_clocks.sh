#!/bin/bash
# tools/lab/r5_clocks.sh: the DEFLATE launch's two speeds on one box -- launch time next to the clocks and the power rocm-smi reports, every few seconds for two minutes
cd "$(dirname "$0")/../.."
[ -f /tmp/lab50.bam ] || bin/synth_bam write /tmp/lab50.bam 50000000 --seed 1 > /dev/null
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -iE "sclk|mclk|fclk|socclk|power|Temperature \(Sensor (junction|memory)" | head -12
for i in $(seq 1 ${N:-24}); do
  ms=$(tools/lab/bin/coop_lab_cur /tmp/lab50.bam 8 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms'], r['median_ms'])")
  smi=$(rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -iE "sclk|mclk|fclk|Average Graphics Package Power|Current Socket|junction" | sed 's/.*: *//; s/GPU\[0\]//' | tr '\n' ' ' | cut -c1-200)
  echo "$i: $ms | $smi"
done
