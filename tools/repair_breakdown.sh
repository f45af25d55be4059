#!/bin/bash
# round 5: per-kernel times of the repair chain on the three inputs of bench.py's repair_spread (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/rb
run() {   # name, bench flags ("shapes <mode>": tools/bench_shapes.py <mode> instead of bench.py)
  if [ "${2%% *}" = shapes ]; then
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rb/$1 -- python3 tools/bench_shapes.py ${2#shapes } > gpurun_out/rb/$1.log 2>&1
  else
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rb/$1 -- python3 bench.py --steps 30 --warmup 10 --cpu-frames 0 --single-stream 0 --extras 0 $2 > gpurun_out/rb/$1.log 2>&1
  fi
  python3 tools/summarize_rocprof.py gpurun_out/rb/$1 gpurun_out/rb/$1.csv > /dev/null
  echo "== $1: bench.py $2"
  python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/rb/$1.csv")):
    if "non-mca" in r["Name"] or "k_bf_table" in r["Name"] or not r["AverageNs"]: continue
    n=r["Name"]; n=n[n.find("k_"):][:60]
    print("%-62s %5s launches %9.1f us" % (n, r["Calls"], float(r["AverageNs"])/1e3))
PY
  rm -rf gpurun_out/rb/$1
}
run bench_8x4096 ""
run shape_128x256 "--arrays 128 --frames 256"
run single_1x4096 "--arrays 1 --frames 4096"
run clustered_8x4096 "shapes s1"
