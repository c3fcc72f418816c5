#!/bin/bash
# GPU-box script: SQ counters of k_inflate_ring alone (and of round 1's kernel for reference) on the 50 M-read bench file.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/ring
bin/synth_bam write /tmp/s50.bam 50000000 --threads 64 > /dev/null 2>&1
R=$PWD; export TMPDIR=/tmp; cd /tmp
for mode in ring r1; do
  if [ $mode = r1 ]; then export REGTOOLS_AMD_INFLATE_R1=1; else unset REGTOOLS_AMD_INFLATE_R1; fi
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" \
             "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/ring/pmc_${mode}_g$i -o p -- $R/tools/lab/bin/inflate_lab_base /tmp/s50.bam 1 > $R/gpurun_out/ring/pmc_${mode}_g$i.log 2>&1
  done
done
cd $R
python3 - <<PY
import csv,glob,collections
for mode in ("ring","r1"):
    agg=collections.defaultdict(list)
    for f in glob.glob("gpurun_out/ring/pmc_%s_g*/**/*counter_collection.csv"%mode,recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_inflate" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==",mode)
    for k,v in sorted(agg.items()): print("%-40s n=%d mean=%.5g"%(k,len(v),sum(v)/len(v)))
PY
