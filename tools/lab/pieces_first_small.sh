for pc in "33,67" "10,40" "15,50" "20,55" "8,30" "25,60" "5,35"; do
  echo -n "pieces $pc: "; REGTOOLS_AMD_PIECES=$pc python bench.py --no-extras --no-cpu-baseline --host-only --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(round(r['ms_per_step'],2))"
done
