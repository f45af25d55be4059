#!/bin/bash
# round 5: per-kernel times of the adaptive 16-microphone call (8 x 2048 frames, one source per array): rocprofv3 kernel trace
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/m16
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/m16/raw -- python3 tools/bench_shapes.py m16a > gpurun_out/m16/run.log 2>&1
grep "^M=" gpurun_out/m16/run.log
python3 tools/summarize_rocprof.py gpurun_out/m16/raw gpurun_out/m16/kernels.csv > /dev/null
python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/m16/kernels.csv")):
    if "non-mca" in r["Name"] or "k_bf_table" in r["Name"] or not r["AverageNs"]: continue
    n=r["Name"]; n=n[n.find("k_"):][:70]
    print("%-72s %5s launches %9.1f us" % (n, r["Calls"], float(r["AverageNs"])/1e3))
PY
rm -rf gpurun_out/m16/raw
