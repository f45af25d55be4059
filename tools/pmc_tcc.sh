cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_tcc; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $out/p1 -- python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --single-stream 0 --extras 0 > $out/p1.log 2>&1
python3 - <<PY
import csv,glob,collections
for f in glob.glob("$out/p1/**/*counter_collection.csv",recursive=True):
    agg=collections.defaultdict(float); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "mca" not in k: continue
        agg[(k[:60],r["Counter_Name"])]+=float(r["Counter_Value"]); cnt[(k[:60],r["Counter_Name"])]+=1
    for (k,c),v in sorted(agg.items()): print("%-62s %-14s %14.0f"%(k,c,v/cnt[(k,c)]))
PY
tail -3 $out/p1.log | cut -c1-200
