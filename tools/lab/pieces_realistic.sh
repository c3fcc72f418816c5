# the upload pieces for the realistic payload (PCIe 54 ms, inflate 52 ms, a member's own chain ~29 ms): does a smaller last piece end sooner?
for P in 3 "40,80" "45,88" 4 5 6; do
  echo "PIECES=$P"; REGTOOLS_AMD_PIECES=$P timeout 300 python bench.py --realistic --host-only --no-extras --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done
