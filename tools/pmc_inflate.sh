#!/bin/bash
# usage: tools/pmc_inflate.sh <outdir> <forms> [extra inflate_bench.py args]
# rocprofv3 --pmc passes (counters only, one group per run) over tools/inflate_bench.py: the DEFLATE kernel alone on the bench file.
OUT=$1; FORMS=${2:-1,4}; shift 2
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/$OUT; cd /tmp
python $R/tools/inflate_bench.py --forms $FORMS --bam /tmp/pmc_bench.bam --reps 1 "$@" > $R/$OUT/plain.log 2>&1
i=0
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $R/$OUT/pmc$i -o p -- python $R/tools/inflate_bench.py --forms $FORMS --bam /tmp/pmc_bench.bam --reps 1 "$@" > $R/$OUT/pmc$i.log 2>&1
done
cd $R
python3 - <<PY
import csv,glob,collections,json
res=collections.defaultdict(dict)
for d in sorted(glob.glob("$OUT/pmc*/")):
    for f in glob.glob(d+"**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0]
            if "k_inflate" in k: res[k][r["Counter_Name"]]=float(r["Counter_Value"])
for k,v in res.items(): print(k, json.dumps(v))
json.dump(res, open("$OUT/pmc_inflate.json","w"), indent=1)
PY
