mkdir -p gpurun_out/r2
for P in 3 4 5 6 7; do
  echo "PIECES=$P"; REGTOOLS_AMD_PIECES=$P timeout 200 python bench.py --host-only --no-extras --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('stage_ms'))"
done
echo "HWQ=8"
for P in 3 6 7; do
  echo "PIECES=$P"; GPU_MAX_HW_QUEUES=8 REGTOOLS_AMD_PIECES=$P timeout 200 python bench.py --host-only --no-extras --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('stage_ms'))"
done
