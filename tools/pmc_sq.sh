#!/bin/bash
# SQ counters of every mca:: kernel (rocprofv3 PMC, two passes of up to 8 counters, --kernel-trace only).
# usage: tools/pmc_sq.sh <precision> <outdir> ["more bench.py flags", e.g. "--config mvdr"]
#        PMC_CMD="python3 tools/bench_shapes.py n2048" tools/pmc_sq.sh n2048 <outdir>     (another program under the same passes; <precision> only names the file)
# Also keeps every kernel's average duration UNDER the counters of pass 1 (kernel trace of the same run): SQ_BUSY_CYCLES / 32 / duration is
# the shader clock the issue-rate roofline of bench.py is priced with.
prec=${1:-fp16x3}; out=${2:-gpurun_out/pmc_sq}; extra=${3:-}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $out
P1="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU"
P2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
i=0
for set in "$P1" "$P2"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- ${PMC_CMD:-python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --single-stream 0 --extras 0 --precision $prec $extra} > $out/p$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections,json
res=collections.defaultdict(dict)
for p in ("p1","p2"):
    for f in glob.glob("$out/%s/**/*counter_collection.csv"%p,recursive=True):
        agg=collections.defaultdict(float); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "mca" not in k: continue
            agg[(k,r["Counter_Name"])]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
        for (k,c),v in agg.items():
            res[k][c]=v/cnt[(k,c)]
            res[k]['dispatches_in_pass']=cnt[(k,c)]; res[k]['steps_in_pass']=3      # (--steps 2 --warmup 1)
for f in glob.glob("$out/p1/**/*kernel_trace.csv",recursive=True):
    dur=collections.defaultdict(float); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "mca" not in k: continue
        dur[k]+=float(r["End_Timestamp"])-float(r["Start_Timestamp"]); n[k]+=1
    for k in dur:
        res[k]["avg_duration_ns_under_pmc"]=dur[k]/n[k]
        if "SQ_BUSY_CYCLES" in res[k]: res[k]["clock_ghz"]=res[k]["SQ_BUSY_CYCLES"]/32.0/(dur[k]/n[k])
json.dump(res,open("$out/sq_$prec.json","w"),indent=1)
for k,v in res.items():
    print(k[:70])
    for c,x in sorted(v.items()): print("   %-26s %14.3f"%(c,x))
PY
# the raw rocprofv3 output stays on the box (gpurun merges at most 64 MiB back): the summaries above are what is kept
rm -rf $out/p1 $out/p2
