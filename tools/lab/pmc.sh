#!/bin/bash
# tools/lab/pmc.sh OUTDIR FILE.bam [variant]  -- hardware counters of k_inflate alone (inflate_lab), one --pmc pass per group.
# FETCH_SIZE / WRITE_SIZE are collected by tools/pmc_traffic.sh on the bench command instead (those passes are slow on this binary).
OUT=$1; BAM=$2; V=${3:-base}; R=$PWD; export TMPDIR=/tmp; mkdir -p $R/$OUT; cd /tmp
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TA_TA_BUSY_sum TA_FLAT_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" "TCC_WRITEBACK_sum TCC_READ_sum TCC_WRITE_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/$OUT/g$i -o p -- $R/tools/lab/bin/inflate_lab_$V $BAM 1 > $R/$OUT/g$i.log 2>&1
done
cd $R
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/g*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_inflate" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print("%-40s n=%d mean=%.5g"%(k,len(v),sum(v)/len(v)))
PY
