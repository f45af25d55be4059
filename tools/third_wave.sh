#!/bin/bash
# round 5: the same analysis kernel at 3 / 2 / 1 waves per SIMD (tools/third_wave.py), times and SQ counters
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export MCA_HIP_LIB=$GRAFT_REPO_ROOT/abtest/lib_measure.so
mkdir -p gpurun_out/tw
for pad in 0 10 40; do
  MCA_HIP_SPW_LDS_PAD=$pad python tools/third_wave.py 2>&1 | grep "LDS pad"
  THIRD_WAVE_STEPS=3 MCA_HIP_SPW_LDS_PAD=$pad timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d gpurun_out/tw/p$pad -- python3 tools/third_wave.py > gpurun_out/tw/p$pad.log 2>&1
  python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(float); cnt=collections.Counter()
for f in glob.glob("gpurun_out/tw/p$pad/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_stft_phat_wave" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
v={k:agg[k]/cnt[k] for k in agg}
if v:
    busy=v["SQ_BUSY_CYCLES"]/32
    print("   counters per launch: waves %d  SQ_INSTS_VALU %.1f M  VALU busy %.1f %% of %.0f k busy cycles  (wave-cycles / busy = %.2f waves per SIMD resident on average)  LDS wait %.1f %% of wave cycles" % (
        v.get("SQ_WAVES",0), v["SQ_INSTS_VALU"]/1e6, 100*v["SQ_ACTIVE_INST_VALU"]*4/1024/busy, busy/1e3, v["SQ_WAVE_CYCLES"]*4/1024/busy, 100*v["SQ_WAIT_INST_LDS"]/v["SQ_WAVE_CYCLES"]))
PY
  rm -rf gpurun_out/tw/p$pad
done
