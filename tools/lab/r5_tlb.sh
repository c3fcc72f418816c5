#!/bin/bash
# tools/lab/r5_tlb.sh: are the DEFLATE launch's two speeds address translation?  Fresh processes one after the other, each under rocprofv3 with the L1 translation
# counters; kernel duration from the same run's kernel trace.
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r5/tlb; mkdir -p $O; export TMPDIR=/tmp
[ -f /tmp/lab50.bam ] || bin/synth_bam write /tmp/lab50.bam 50000000 --seed 1 > /dev/null
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -o "[A-Z_0-9]*UTCL[A-Z_0-9a-z]*" | sort -u | head -40 > $O/utcl_counters.txt
for i in $(seq 1 ${N:-8}); do
  rocprofv3 --pmc ${CTRS:-TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum} --kernel-trace --output-format csv -d $O/p$i -o p -- $R/tools/lab/bin/coop_lab_ptr /tmp/lab50.bam 2 > $O/p$i.json 2> $O/p$i.err
  python3 - $O/p$i <<'PY'
import csv,glob,sys,collections
d=sys.argv[1]
c=collections.defaultdict(list)
for f in glob.glob(d+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_inflate_coop" in r["Kernel_Name"]: c[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur=[]
for f in glob.glob(d+"/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_inflate_coop" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
print(d.split("/")[-1], "ms", [round(x,2) for x in dur], {k:sum(v)/len(v) for k,v in c.items()})
PY
done
