#!/bin/bash
# usage: tools_pmc.sh <outdir> <reads>   -- rocprofv3 PMC passes over one bench run (counters only, own runs)
OUT=$1; READS=${2:-5000000}
R=$PWD; export TMPDIR=/tmp; cd /tmp
i=0
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $R/$OUT/pmc$i -o p -- python $R/bench.py --reads $READS --steps 1 --warmup 1 --no-cpu-baseline > $R/$OUT/pmc$i.log 2>&1
done
cd $R
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob("$OUT/pmc*/")):
    for f in glob.glob(d+"**/*counter_collection.csv",recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
        for k in agg:
            if "inflate" in k or "decode" in k or "seg_" in k:
                print(k, {c:(v/cnt[(k,c)]) for c,v in agg[k].items()}, "dispatches", max(cnt[(k,c)] for c in agg[k]))
PY
