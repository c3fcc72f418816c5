#!/bin/bash
# tools/cli_wall.sh [reads]: cold-process wall time of `regtools-amd junctions extract` next to the reference's, same file (both from the page cache)
R=${1:-50000000}
[ -f /tmp/b$R.bam ] || bin/synth_bam write /tmp/b$R.bam $R --seed 1 >/dev/null
for i in 1 2 3; do
  s=$(date +%s.%N); REGTOOLS_AMD_STATS=1 REGTOOLS_AMD_TRACE=${TRACE:-0} bin/regtools-amd junctions extract -s XS -o /tmp/out.bed /tmp/b$R.bam 2> /tmp/cli.err; e=$(date +%s.%N)
  echo "cli wall $(echo "$e - $s" | bc) s"; grep -v "^Minimum\|^Maximum\|^Alignment\|^Output\|^$" /tmp/cli.err | tail -${TAILN:-3}
done
if [ -x oracle/_ref/regtools_ref ]; then
  s=$(date +%s.%N); oracle/_ref/regtools_ref junctions extract -s XS -o /tmp/ref.bed /tmp/b$R.bam 2>/dev/null; e=$(date +%s.%N)
  echo "reference wall $(echo "$e - $s" | bc) s"; cmp /tmp/out.bed /tmp/ref.bed && echo identical
fi
