cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
for p in fp16x3 fp16 fp32; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/rocprof_$p -- python3 bench.py --steps 100 --warmup 20 --cpu-frames 0 --precision $p > gpurun_out/final/rocprof_$p.log 2>&1
  python3 tools/summarize_rocprof.py gpurun_out/final/rocprof_$p gpurun_out/final/kernel_stats_$p.csv > /dev/null
done
timeout 900 python bench.py > gpurun_out/final/bench_fp16x3.log 2>&1
timeout 900 python bench.py --precision fp16 --cpu-frames 0 > gpurun_out/final/bench_fp16.log 2>&1
timeout 900 python bench.py --precision fp32 --cpu-frames 0 > gpurun_out/final/bench_fp32.log 2>&1
# BASELINE configs[3]: the MVDR path (256 streams x 64 frames, 16 microphones)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/rocprof_mvdr -- python3 tools/bench_mvdr_dev.py --check 0 --steps 10 > gpurun_out/final/rocprof_mvdr.log 2>&1
python3 tools/summarize_rocprof.py gpurun_out/final/rocprof_mvdr gpurun_out/final/kernel_stats_mvdr.csv > /dev/null
timeout 300 python tools/bench_mvdr_dev.py > gpurun_out/final/bench_mvdr.log 2>&1
tail -c 600 gpurun_out/final/bench_fp16x3.log
cat gpurun_out/final/kernel_stats_fp16x3.csv
