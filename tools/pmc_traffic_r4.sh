#!/bin/bash
# (round 4 script, kept as the record of how profiles/r04_pmc_traffic.json was made: REGTOOLS_AMD_INFLATE_TUNE / _PAIRS left the product in round 5 -- bench.py measures the traffic in its own run now)
# tools/pmc_traffic_r4.sh OUTDIR: FETCH_SIZE / WRITE_SIZE of the DEFLATE launch alone (tools/lab/bin/coop_lab_cur = the product's kernels + the lab's
# timing main, tools/lab/build_cur.sh cur) on the three bench payloads, in the options the pipeline picks for each (kernels.h inflate_plan_for);
# separate --pmc passes, as the guide prescribes.  Writes OUTDIR/r04_pmc_traffic.json (copied to profiles/ by hand).
OUT=$1
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/$OUT
[ -f /tmp/lab50.bam ] || bin/synth_bam write /tmp/lab50.bam 50000000 --seed 1 > /dev/null
[ -f /tmp/labr50.bam ] || bin/synth_bam write /tmp/labr50.bam 50000000 --seed 1 --realistic > /dev/null
[ -f /tmp/labl10.bam ] || bin/synth_bam write /tmp/labl10.bam 10000000 --seed 1 --shape long > /dev/null
cd /tmp
for w in default:lab50: realistic:labr50: long10M:labl10:"REGTOOLS_AMD_INFLATE_TUNE=2 REGTOOLS_AMD_INFLATE_PAIRS=4"; do
  key=${w%%:*}; rest=${w#*:}; f=${rest%%:*}; envs=${rest#*:}
  env $envs $R/tools/lab/bin/coop_lab_cur /tmp/$f.bam 3 > $R/$OUT/$key.lab.json 2>/dev/null
  stat -c %s /tmp/$f.bam > $R/$OUT/$key.size
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
    d=$(echo $c | tr ' ' '_')
    env $envs timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/${key}_$d -o p -- $R/tools/lab/bin/coop_lab_cur /tmp/$f.bam 1 > $R/$OUT/${key}_$d.log 2>&1
  done
done
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {}
for key in ("default", "realistic", "long10M"):
    lab = json.loads(open("%s/%s.lab.json" % (out, key)).read())
    c = collections.defaultdict(list); kern = None
    for f in glob.glob("%s/%s_*/**/*counter_collection.csv" % (out, key), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_inflate" in r["Kernel_Name"]:
                c[r["Counter_Name"]].append(float(r["Counter_Value"])); kern = r["Kernel_Name"].split("(")[0].replace("void ", "")
    v = {k: sum(x) / len(x) for k, x in c.items()}
    # algorithmic bytes = compressed payload bytes + inflated bytes of the members launched (bench.py uses the table's compressed_bytes = the file's length)
    fsize = int(open("%s/%s.size" % (out, key)).read())
    res[key] = dict(kernel=kern, algorithmic_bytes=fsize + lab["inflated"], compressed_bytes=fsize, lab_ms=lab["ms"], members=lab["members"], inflated_bytes=lab["inflated"], FETCH_SIZE_KiB=v.get("FETCH_SIZE"), WRITE_SIZE_KiB=v.get("WRITE_SIZE"), counters=v)
json.dump(res, open(out + "/r04_pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: {x: y for x, y in v.items() if x != "counters"} for k, v in res.items()}, indent=1))
PY
