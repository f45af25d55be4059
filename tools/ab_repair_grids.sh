#!/bin/bash
# round 5: the fixed grids of the three repair launches (list-mode analysis 512, k_srp_cand 512, k_scan_repick 256 workgroups) against smaller ones;
# MEASURE build.  Per-kernel times from rocprofv3 on the bench input and the 1 x 4096 call.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export MCA_HIP_LIB=$GRAFT_REPO_ROOT/abtest/lib_measure.so
for g in "512 512 256" "512 512 64"; do
  set -- $g
  for shape in "" "--arrays 1 --frames 4096"; do
    MCA_HIP_LIST_GRID=$1 MCA_HIP_CAND_GRID=$2 MCA_HIP_REPICK_GRID=$3 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rg -- python3 bench.py --steps 30 --warmup 10 --cpu-frames 0 --single-stream 0 --extras 0 $shape > gpurun_out/rg.log 2>&1
    python3 tools/summarize_rocprof.py gpurun_out/rg gpurun_out/rg.csv > /dev/null
    python3 - "$g" "$shape" <<PY
import csv,sys
r={}
for row in csv.DictReader(open("gpurun_out/rg.csv")):
    n=row["Name"]
    if not row["AverageNs"]: continue
    for k,t in (("Lb1ELb0ELb0ELb0","list analysis"),("k_srp_cand","cand"),("k_scan_repick","repick")):
        if k in n: r[t]=float(row["AverageNs"])/1e3
print("grids %-12s %-26s %s  sum %.1f us" % (sys.argv[1], sys.argv[2] or "8 x 4096", {k: round(v,1) for k,v in r.items()}, sum(r.values())))
PY
    rm -rf gpurun_out/rg
  done
done
