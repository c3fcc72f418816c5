#!/bin/bash
# usage: tools/ktrace.sh <outdir> [bench args...]  -- rocprofv3 kernel trace of one bench run, per-kernel summary
OUT=$1; shift
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/$OUT; cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT -o kt -- python $R/bench.py --no-cpu-baseline "$@" > $R/$OUT/bench.log 2>&1
cd $R
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/**/kt_kernel_trace.csv",recursive=True)[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
tot=sum(sum(v) for v in agg.values())
print("%-34s %6s %12s %10s %10s %10s %6s"%("kernel","calls","total_us","avg_us","min_us","max_us","pct"))
for k,v in sorted(agg.items(),key=lambda kv:-sum(kv[1])):
    print("%-34s %6d %12.1f %10.1f %10.1f %10.1f %6.2f"%(k[:34],len(v),sum(v),sum(v)/len(v),min(v),max(v),100*sum(v)/tot))
PY
