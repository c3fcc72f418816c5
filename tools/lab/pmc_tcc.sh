#!/bin/bash
# tools/lab/pmc_tcc.sh OUTDIR variant...: L2 <-> fabric request counters of the lab variants (who misses: reads or writes)
OUT=$1; shift
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/$OUT
[ -f /tmp/lab50.bam ] || bin/synth_bam write /tmp/lab50.bam 50000000 --seed 1 > /dev/null
cd /tmp
for v in "$@"; do
  i=0
  for pmc in "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum" "TCC_READ_sum TCC_WRITE_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_TAG_STALL_sum"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $R/$OUT/${v}_$i -o p -- $R/tools/lab/bin/coop_lab_$v /tmp/lab50.bam 1 > $R/$OUT/${v}_$i.log 2>&1
  done
done
cd $R
python3 - "$OUT" "$@" <<'PY'
import csv,glob,sys,json,collections
out=sys.argv[1]
for v in sys.argv[2:]:
    res=collections.defaultdict(list)
    for f in glob.glob("%s/%s_*/**/*counter_collection.csv"%(out,v),recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_inflate" in r["Kernel_Name"]: res[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(v, json.dumps({k:sum(x)/len(x) for k,x in sorted(res.items())}))
PY
