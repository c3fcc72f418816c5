#!/bin/bash
# tools/lab/r5_arena_counters.sh: WHICH hardware counter follows the DEFLATE launch's time from arena to arena?  tools/lab/bin/coop_lab_marena (eight hipMalloc arenas of one
# process, the same launch into each) under rocprofv3 --pmc, one counter set per pass; per dispatch: duration (kernel trace) against the counters; the correlation over dispatches.
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r5/arena_counters; mkdir -p $O; export TMPDIR=/tmp
[ -f /tmp/lab50.bam ] || bin/synth_bam write /tmp/lab50.bam 50000000 --seed 1 > /dev/null
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "\b\(TCC\|TCP\|GRBM\|TCA\|MALL\|EA\|UTCL2\|GCVML2\|VML2\|ATC\)_[A-Za-z0-9_]*" | sort -u > $O/counter_names.txt
i=0
while read -r pmc; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $O/pass$i -o p -- $R/tools/lab/bin/${BIN:-coop_lab_marena} /tmp/lab50.bam 1 > $O/pass$i.log 2>&1 || echo "pass $i ($pmc) failed: $(tail -2 $O/pass$i.log | tr '\n' ' ')"
done < ${SETS_FILE:-$R/tools/lab/r5_arena_counters.sets2}
cd $R
python3 - $O <<'PY'
import csv, glob, sys, collections, math
O = sys.argv[1]
for d in sorted(glob.glob(O + "/pass*/")):
    dur = {}
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_inflate_coop" in r["Kernel_Name"]: dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    ctr = collections.defaultdict(dict)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_inflate_coop" in r["Kernel_Name"]: ctr[r["Counter_Name"]][r["Dispatch_Id"]] = ctr[r["Counter_Name"]].get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    ids = sorted(dur, key=int)
    if not ids: print(d, "no dispatches"); continue
    t = [dur[i] for i in ids]
    print("%s: %d dispatches, %.2f .. %.2f ms" % (d.rstrip("/").split("/")[-1], len(ids), min(t), max(t)))
    for name, vals in sorted(ctr.items()):
        v = [vals.get(i, 0.0) for i in ids]
        mt, mv = sum(t) / len(t), sum(v) / len(v)
        st, sv = math.sqrt(sum((x - mt) ** 2 for x in t)), math.sqrt(sum((x - mv) ** 2 for x in v))
        corr = sum((a - mt) * (b - mv) for a, b in zip(t, v)) / (st * sv) if st > 0 and sv > 0 else float("nan")
        fast = [b for a, b in zip(t, v) if a <= sorted(t)[len(t) // 4]]; slow = [b for a, b in zip(t, v) if a >= sorted(t)[3 * len(t) // 4]]
        print("   %-52s corr with time %+.3f   mean %.4g   fastest quarter %.4g   slowest quarter %.4g" % (name, corr, mv, sum(fast) / len(fast), sum(slow) / len(slow)))
PY
