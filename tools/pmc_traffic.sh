#!/bin/bash
# HBM traffic of the dominant kernel (rgx::k_inflate) at the bench configuration + calibration of the counters on a kernel
# with the same access pattern and a known byte count (tools/ubench_unaligned.hip).  Separate --pmc passes (guide).
OUT=$1; R=$PWD; export TMPDIR=/tmp; mkdir -p $R/$OUT
hipcc --offload-arch=gfx950 -O3 $R/tools/ubench_unaligned.hip -o /tmp/ub 2>/dev/null
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/cal_$c -o p -- /tmp/ub > $R/$OUT/cal_$c.log 2>&1
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/bench_$c -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $R/$OUT/bench_$c.log 2>&1
done
cd $R
python3 - <<PY
import csv,glob,json
def grab(d,kern,mingrid=0):
    out=[]
    for f in glob.glob(d+"/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"] and int(r["Grid_Size"])>=mingrid: out.append(float(r["Counter_Value"]))
    return out
res={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    res["cal_"+c]=grab("$OUT/cal_"+c,"k_copy")
    res["inflate_"+c]=grab("$OUT/bench_"+c,"k_inflate",100000)      # the whole-file launches (the overlapped pieces have smaller grids)
res["cal_known_bytes_each_way"]=4000*16*256*448
json.dump(res,open("$OUT/pmc_traffic.json","w"),indent=1)
print(json.dumps(res))
PY
