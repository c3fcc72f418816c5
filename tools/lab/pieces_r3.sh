# round 3: the upload piece sweep again, with k_inflate_coop (the round-2 sweep used k_inflate)
for P in 3 4 5 6 "12,37,68" "20,55" "10,30,60"; do
  echo "PIECES=$P"; REGTOOLS_AMD_PIECES=$P timeout 200 python bench.py --host-only --no-extras --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done
