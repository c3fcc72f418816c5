#!/bin/bash
# tools/lab/bisect_r4.sh OUTDIR VARIANTS...: time on the three payloads + one SQ counter pass on the bench file, per lab variant
OUT=$1; shift
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/$OUT
[ -f /tmp/lab50.bam ] || bin/synth_bam write /tmp/lab50.bam 50000000 --seed 1 > /dev/null
[ -f /tmp/labr50.bam ] || bin/synth_bam write /tmp/labr50.bam 50000000 --seed 1 --realistic > /dev/null
[ -f /tmp/labl10.bam ] || bin/synth_bam write /tmp/labl10.bam 10000000 --seed 1 --shape long > /dev/null
for v in "$@"; do
  for f in lab50 labr50 labl10; do
    echo -n "$v $f: "; $R/tools/lab/bin/coop_lab_$v /tmp/$f.bam 4 2> /tmp/trips.txt | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms'], r['fnv64_first64MiB'], r['bad_member'], end=' ')"; cat /tmp/trips.txt | tr '\n' ' '; echo
  done
  ( cd /tmp && timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/$OUT/${v} -o p -- $R/tools/lab/bin/coop_lab_$v /tmp/lab50.bam 1 > $R/$OUT/${v}.log 2>&1 )
done
cd $R
python3 - "$OUT" "$@" <<'PY'
import csv,glob,sys,json,collections
out=sys.argv[1]
for v in sys.argv[2:]:
    res=collections.defaultdict(list)
    for f in glob.glob("%s/%s/**/*counter_collection.csv"%(out,v),recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_inflate" in r["Kernel_Name"]: res[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(v, json.dumps({k:round(sum(x)/len(x)/1e6,1) for k,x in sorted(res.items())}))
PY
