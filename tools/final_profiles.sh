# Round profiles: the driver's bench command, bench lines per precision, rocprofv3 kernel stats, PMC traffic / SQ counters, shapes.
# usage (GPU box, from the repo root): bash tools/final_profiles.sh [a|b|c] (parts that each fit one 20-minute gpurun call; default a and b) ; then python tools/collect_profiles.py r06 here
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
PART=${1:-ab}
if [[ $PART == *a* ]]; then
# the driver's exact command, and the two short warm-ups VERDICT r3 asked about
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench_driver_cmd.log 2>&1
grep "^{" gpurun_out/final/bench_driver_cmd.log | tail -1 > gpurun_out/final/bench_driver_cmd.json      # (the compact line, as the driver sees it)
cp gpurun_out/bench_detail.json gpurun_out/final/bench_driver_cmd_detail.json                                # (... and the full record of the same run)
for w in 0 1; do
  timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup $w --cpu-frames 0 --single-stream 0 --extras 0 2> /dev/null | grep "^{" | tail -1 > gpurun_out/final/bench_warmup$w.json
done
# rocprofv3 kernel trace of the driver's command (the configs[1] / configs[3] / spread extras off: they launch the same kernels on other shapes)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/rocprof_driver -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --single-stream 0 --extras 0 > gpurun_out/final/rocprof_driver.log 2>&1
python3 tools/summarize_rocprof.py gpurun_out/final/rocprof_driver gpurun_out/final/kernel_stats_driver_cmd.csv > /dev/null; rm -rf gpurun_out/final/rocprof_driver
for p in adaptive fp16x3; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/rocprof_$p -- python3 bench.py --steps 100 --warmup 20 --cpu-frames 0 --single-stream 0 --extras 0 --precision $p > gpurun_out/final/rocprof_$p.log 2>&1
  python3 tools/summarize_rocprof.py gpurun_out/final/rocprof_$p gpurun_out/final/kernel_stats_$p.csv > /dev/null; rm -rf gpurun_out/final/rocprof_$p
done
timeout 900 python bench.py --full > gpurun_out/final/bench_adaptive.log 2>&1
timeout 300 python bench.py --full --arrays 128 --frames 256 --cpu-frames 0 --single-stream 0 --extras 0 > gpurun_out/final/bench_128x256.log 2>&1
grep "^{" gpurun_out/final/bench_128x256.log | tail -1 > gpurun_out/final/bench_128x256.json
timeout 900 python bench.py --full --precision fp16x3 --cpu-frames 0 --extras 0 > gpurun_out/final/bench_fp16x3.log 2>&1
timeout 900 python bench.py --full --precision fp16 --cpu-frames 0 --extras 0 > gpurun_out/final/bench_fp16.log 2>&1
timeout 900 python bench.py --full --precision fp32 --cpu-frames 0 --extras 0 --steps 30 --warmup 5 > gpurun_out/final/bench_fp32.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/rocprof_mvdr -- python3 bench.py --config mvdr --steps 20 --warmup 5 --cpu-frames 0 > gpurun_out/final/rocprof_mvdr.log 2>&1
python3 tools/summarize_rocprof.py gpurun_out/final/rocprof_mvdr gpurun_out/final/kernel_stats_mvdr.csv > /dev/null; rm -rf gpurun_out/final/rocprof_mvdr
timeout 300 python bench.py --full --config mvdr > gpurun_out/final/bench_mvdr.log 2>&1
fi
if [[ $PART == *b* ]]; then
bash tools/pmc_traffic.sh adaptive gpurun_out/pmc_traffic_adaptive > gpurun_out/final/pmc_traffic.log 2>&1
bash tools/pmc_sq.sh adaptive gpurun_out/pmc_sq > gpurun_out/final/pmc_sq.log 2>&1
timeout 600 python tools/precision_report.py > gpurun_out/final/precision_report.json 2> gpurun_out/final/precision_report.log
timeout 300 python tools/host_path_rate.py > gpurun_out/final/host_path.log 2>&1
python tools/bench_fallback.py 2>&1 | grep -v amdgpu.ids > gpurun_out/final/fallback.log
(python tools/bench_shapes.py; python tools/bench_shapes.py m16; python tools/bench_shapes.py sources; python tools/bench_shapes.py gate; python tools/bench_shapes.py n2048; python tools/bench_shapes.py n512; echo "--- the any-length kernels at 2048-sample frames (MCA_HIP_NO_N2048=1):"; MCA_HIP_NO_N2048=1 python tools/bench_shapes.py n2048) 2>&1 | grep -v amdgpu.ids > gpurun_out/final/shapes.log
timeout 300 python tools/stream_latency.py 2>&1 | grep -v amdgpu.ids > gpurun_out/final/stream_latency.log
fi
if [[ $PART == *c* ]]; then
# round 5: the MVDR kernels' PMC passes, the adaptive mode against FP16X3 at sizes the oracle cannot reach, the per-kernel
# breakdown of the repair chain on three shapes, the kernel timeline of one step
bash tools/r05_profiles_mvdr.sh > gpurun_out/final/pmc_mvdr.log 2>&1
timeout 900 python tools/adaptive_check.py 40 2026 > gpurun_out/final/adaptive_check.json 2> gpurun_out/final/adaptive_check.log
bash tools/repair_breakdown.sh > gpurun_out/final/repair_breakdown.log 2>&1
bash tools/timeline.sh > gpurun_out/final/timeline.log 2>&1
fi
tail -c 300 gpurun_out/final/bench_driver_cmd.json
cat gpurun_out/final/kernel_stats_driver_cmd.csv
