cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "MCA_HIP_ADAPT_CAND=0" "MCA_HIP_ADAPT_CAND=0 MCA_HIP_ADAPT_LAZY=0" "MCA_HIP_ADAPT_CAND=1"; do
  export $v
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sp -- python3 bench.py --steps 30 --warmup 10 --cpu-frames 0 --single-stream 0 --extras 0 --arrays 128 --frames 256 > gpurun_out/sp.log 2>&1
  python3 tools/summarize_rocprof.py gpurun_out/sp gpurun_out/sp.csv > /dev/null
  echo "== $v"; grep -E "scan_pick|repick|cand|repair|patch" gpurun_out/sp.csv | cut -d, -f1,4 | cut -c1-90
  rm -rf gpurun_out/sp; unset MCA_HIP_ADAPT_CAND MCA_HIP_ADAPT_LAZY
done
