#!/bin/bash
OUT=$1; READS=${2:-50000000}
R=$PWD; export TMPDIR=/tmp; cd /tmp
i=0
for pmc in "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TA_TA_BUSY_sum TA_FLAT_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $R/$OUT/q$i -o p -- python $R/bench.py --reads $READS --steps 1 --warmup 0 --no-cpu-baseline > $R/$OUT/q$i.log 2>&1
done
cd $R
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob("$OUT/q*/")):
    for f in glob.glob(d+"**/*counter_collection.csv",recursive=True):
        rows=list(csv.DictReader(open(f)))
        # keep the biggest k_inflate dispatch only
        best=collections.defaultdict(float)
        for r in rows:
            if "k_inflate" in r["Kernel_Name"] and int(r["Grid_Size"])>100000: best[r["Counter_Name"]]=max(best[r["Counter_Name"]],float(r["Counter_Value"]))
        print(dict(best))
PY
