# A/B of the per-piece record-framing walk (REGTOOLS_AMD_NO_PREWALK=1 = the walk after the whole inflate, as before)
for rep in 1 2; do
for v in 0 1; do
  if [ $v = 1 ]; then export REGTOOLS_AMD_NO_PREWALK=1; else unset REGTOOLS_AMD_NO_PREWALK; fi
  echo "NO_PREWALK=$v"; timeout 200 python bench.py --host-only --no-extras --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('stage_ms'))"
done; done
