#!/bin/bash
# tools/lab/pmc_coop.sh OUTDIR variant...: memory-side counters of the lab variants on the bench file (separate --pmc passes)
OUT=$1; shift
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/$OUT
[ -f /tmp/lab50.bam ] || bin/synth_bam write /tmp/lab50.bam 50000000 --seed 1 > /dev/null
cd /tmp
for v in "$@"; do
  i=0
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM"; do
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
