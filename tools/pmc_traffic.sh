#!/bin/bash
# HBM traffic of every mca:: kernel from rocprofv3 PMC counters, two separate passes (FETCH_SIZE costs 3 of the
# 4 TCC slots, WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots").  usage: tools/pmc_traffic.sh <precision> <outdir>
prec=${1:-fp16x3}; out=${2:-gpurun_out/pmc_traffic}; extra=${3:-}       # extra: more bench.py flags, e.g. "--arrays 128 --frames 256" or "--config mvdr"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $out
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/$ctr -- python3 bench.py --steps 3 --warmup 1 --cpu-frames 0 --single-stream 0 --extras 0 --precision $prec $extra > $out/$ctr.log 2>&1
done
python3 - <<PY
import csv,glob,collections,json
res=collections.defaultdict(dict)
for ctr in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob("$out/%s/**/*counter_collection.csv"%ctr,recursive=True)
    agg=collections.defaultdict(float); cnt=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"]
        if "mca" not in k: continue
        agg[k]+=float(r["Counter_Value"]); cnt[k]+=1
    for k in agg: res[k][ctr]={"sum":agg[k],"dispatches":cnt[k]}
json.dump(res,open("$out/traffic_$prec.json","w"),indent=1)
for k,v in res.items(): print(k[:60], {c:(round(x["sum"]/x["dispatches"],1), x["dispatches"]) for c,x in v.items()})
PY
# the raw rocprofv3 output stays on the box (gpurun merges at most 64 MiB back): the summaries above are what is kept
rm -rf $out/FETCH_SIZE $out/WRITE_SIZE
