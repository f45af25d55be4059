#!/bin/bash
# round 5: start / end of every kernel of one late step of bench.py (rocprofv3 kernel trace), to see what runs beside what
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl/raw -- python3 bench.py --steps 12 --warmup 6 --cpu-frames 0 --single-stream 0 --extras 0 --no-kernel-timing $1 > gpurun_out/tl/run.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/tl/raw/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "mca" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last full step: find the last k_stft_phat_wave (coarse: Lb0ELb0ELb0ELb1) launches
idx=[i for i,r in enumerate(rows) if "k_stft_phat_wave" in r["Kernel_Name"] and "Lb1EEE" in r["Kernel_Name"]]
i0=idx[-3]; i1=idx[-2]
t0=int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1+1]:
    n=r["Kernel_Name"]; n=n[n.find("k_"):][:40]
    print("%-42s queue %-4s start %8.1f us  dur %7.1f us  end %8.1f" % (n, r.get("Queue_Id","?"), (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, (int(r["End_Timestamp"])-t0)/1e3))
PY
rm -rf gpurun_out/tl/raw
