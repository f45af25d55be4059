#!/bin/bash
# Matrix-core utilisation of the SRP contraction from rocprofv3 PMC counters (--kernel-trace only, one pass):
# SQ_VALU_MFMA_BUSY_CYCLES (cycles a SIMD's matrix pipe is busy, summed over SIMDs: 32 per v_mfma_f32_32x32x16_f16),
# SQ_INSTS_VALU_MFMA_F16, SQ_BUSY_CU_CYCLES and GRBM_GUI_ACTIVE.  usage: tools/pmc_mfma.sh <precision> <outdir>
prec=${1:-fp16x3}; out=${2:-gpurun_out/pmc_mfma}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F16 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/p1 -- python3 bench.py --steps 3 --warmup 1 --cpu-frames 0 --single-stream 0 --extras 0 --precision $prec > $out/p1.log 2>&1
python3 - <<PY
import csv,glob,collections,json
res=collections.defaultdict(dict)
for f in glob.glob("$out/p1/**/*counter_collection.csv",recursive=True):
    agg=collections.defaultdict(float); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "srp_gemm" not in k: continue
        agg[(k,r["Counter_Name"])]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
    for (k,c),v in agg.items(): res[k][c]=v/cnt[(k,c)]
for k,v in res.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
        v["mfma_pipe_utilisation"]=v["SQ_VALU_MFMA_BUSY_CYCLES"]/(v["GRBM_GUI_ACTIVE"]*1024.0)   # 256 CUs x 4 SIMDs
json.dump(res,open("$out/mfma_$prec.json","w"),indent=1)
print(json.dumps(res,indent=1))
PY
