#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_check.sh <tag> [pytest-args]
# runs the GPU parity tests, then bench.py for the three SRP precisions; logs under gpurun_out/<tag>/
tag=${1:-run}; shift
mkdir -p gpurun_out/$tag
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 "$@" 2>&1 | tail -40 > gpurun_out/$tag/tests.log
cat gpurun_out/$tag/tests.log
for p in fp32 fp16x3 fp16; do
  timeout 300 python bench.py --steps 10 --warmup 2 --precision $p --cpu-frames 0 --extras 0 > gpurun_out/$tag/bench_$p.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/$tag/bench_$p.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("$p", round(d["value"]/1e6,2), "Mfps", {k:round(v["avg_ms"],3) for k,v in d["kernels"].items()}, d["roofline"]["kernel"], round(d["roofline"]["frac"],3))
else:
    print("$p FAILED"); print(open("gpurun_out/$tag/bench_$p.log").read()[-1500:])
PY
done
