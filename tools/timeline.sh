#!/bin/bash
# usage: tools/timeline.sh <outdir> [bench args...]  -- rocprofv3 kernel trace of one bench run; start/end of every k_inflate launch and of the
# memory copies of the LAST step, relative to that step's first event (who overlaps whom)
OUT=$1; shift
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/$OUT; cd /tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$OUT -o tl -- python $R/bench.py --no-cpu-baseline --no-live-traffic --host-only "$@" > $R/$OUT/bench.log 2>&1
cd $R
python3 - <<PY
import csv,glob
k=glob.glob("$OUT/**/tl_kernel_trace.csv",recursive=True)[0]
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0][:40],r.get("Queue_Id","?")) for r in csv.DictReader(open(k))]
m=glob.glob("$OUT/**/tl_memory_copy_trace.csv",recursive=True)
for f in m:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+r.get("Direction","?")[:24],"-"))
rows.sort()
infl=[i for i,r in enumerate(rows) if "k_inflate" in r[2]]
# the last step = from the last group of inflate launches
last=infl[-1]; t_end=rows[last][1]
first=[i for i in infl if rows[i][0] > t_end-80e6][0]
t0=min(r[0] for r in rows[max(0,first-40):first+1] if r[0] > rows[first][0]-15e6)
for r in rows:
    if r[0] >= t0 and r[0] <= t_end+12e6 and (r[1]-r[0] > 200e3 or "inflate" in r[2] or "COPY" in r[2]):
        print("%9.3f -> %9.3f ms  (%7.3f)  q=%s  %s"%((r[0]-t0)/1e6,(r[1]-t0)/1e6,(r[1]-r[0])/1e6,r[3],r[2]))
PY
