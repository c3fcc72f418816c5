#!/bin/bash
# GPU-box script: FETCH_SIZE / WRITE_SIZE of every kernel of one headline step (separate --pmc passes, counters only), to set the tail's kernels
# (k_seg_walk, k_decode_seg, k_emit_short, the radix passes) against their algorithmic bytes.  usage: tools/pmc_tail.sh <outdir>
R=$(cd "$(dirname "$0")/.." && pwd); OUT=${1:-gpurun_out/r3/pmc_tail}; mkdir -p $R/$OUT; export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/$c -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $R/$OUT/$c.log 2>&1
done
cd $R
python - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, c), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                agg[r["Kernel_Name"].split("(")[0]][c].append(float(r["Counter_Value"]))
res = {}
for k, d in agg.items():
    # the LAST launch of a kernel is the device-resident pass's (whole-file launches); KiB -> bytes
    res[k] = {c: round(v[-1] * 1024) for c, v in d.items()}
    res[k]["launches"] = max(len(v) for v in d.values())
json.dump(res, open("%s/pmc_tail.json" % out, "w"), indent=1)
for k in sorted(res, key=lambda k: -(res[k].get("FETCH_SIZE", 0) + res[k].get("WRITE_SIZE", 0)))[:14]:
    print("%-60s fetch %7.2f GB  write %7.2f GB  (launches %d)" % (k[:60], res[k].get("FETCH_SIZE", 0) / 1e9, res[k].get("WRITE_SIZE", 0) / 1e9, res[k]["launches"]))
PY
find $R/$OUT -name "*kernel_trace.csv" -size +1M -delete; find $R/$OUT -name "*counter_collection.csv" -size +4M -delete
