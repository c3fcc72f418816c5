#!/bin/bash
# usage: tools/timeline_sustained.sh <outdir> [sustained_ab args...]  -- rocprofv3 kernel + memory-copy trace of tools/sustained_ab.py; the last 75 ms (WINDOW_MS=n: the last n ms) of the
# trace as one timeline: every DEFLATE launch, every copy, every kernel above 150 us, with its queue (which file's context it belongs to)
OUT=$1; shift
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/$OUT; cd /tmp
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$OUT -o tl -- python $R/tools/sustained_ab.py "$@" > $R/$OUT/run.log 2>&1
cd $R
python3 - <<PY
import csv,glob
k=glob.glob("$OUT/**/tl_kernel_trace.csv",recursive=True)[0]
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0].replace("rgx::","")[:44],r.get("Queue_Id","?")) for r in csv.DictReader(open(k))]
for f in glob.glob("$OUT/**/tl_memory_copy_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+r.get("Direction","?")[:24],"-"))
rows.sort()
t_end=max(r[1] for r in rows); t0=t_end-${WINDOW_MS:-75}e6
for r in rows:
    if r[0] >= t0 and (r[1]-r[0] > 150e3 or "inflate" in r[2]):
        print("%9.3f -> %9.3f ms  (%7.3f)  q=%s  %s"%((r[0]-t0)/1e6,(r[1]-t0)/1e6,(r[1]-r[0])/1e6,r[3],r[2]))
PY
