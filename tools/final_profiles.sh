cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
for p in fp16x3 fp16 fp32; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/rocprof_$p -- python3 bench.py --steps 30 --warmup 3 --cpu-frames 0 --precision $p > gpurun_out/final/rocprof_$p.log 2>&1
  python3 tools/summarize_rocprof.py gpurun_out/final/rocprof_$p gpurun_out/final/kernel_stats_$p.csv > /dev/null
done
timeout 900 python bench.py > gpurun_out/final/bench_fp16x3.log 2>&1
timeout 900 python bench.py --precision fp16 --cpu-frames 0 > gpurun_out/final/bench_fp16.log 2>&1
timeout 900 python bench.py --precision fp32 --cpu-frames 0 > gpurun_out/final/bench_fp32.log 2>&1
tail -c 600 gpurun_out/final/bench_fp16x3.log
cat gpurun_out/final/kernel_stats_fp16x3.csv
