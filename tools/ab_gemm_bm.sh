#!/bin/bash
# round 5: the coarse contraction on 128 x 384 tiles / one partial map (shipped) against 256 x 384 tiles / two K halves (MCA_HIP_GEMM_BM256, MEASURE build)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export MCA_HIP_LIB=$GRAFT_REPO_ROOT/abtest/lib_measure.so
run() {
  python bench.py --steps 100 --warmup 20 --cpu-frames 0 --single-stream 0 --extras 0 $2 2> /dev/null | grep "^{" | tail -1 > /tmp/ab.json
  python - "$1" <<PY
import json,sys
d=json.load(open('/tmp/ab.json'))
print('%-44s %6.2f M frames/s  %.4f ms  ' % (sys.argv[1], d['value']/1e6, d['ms_per_step']), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items() if v['launches']})
PY
}
for rep in 1 2 3; do
MCA_HIP_GEMM_BM256=1 run "256 x 384 tiles, two partial maps"
MCA_HIP_GEMM_WN2=1 run "128 x 384 tiles, 4 x 2 waves (32 x 192 each)"
run "128 x 384 tiles, 2 x 4 waves (64 x 96 each)"
done
MCA_HIP_GEMM_BM256=1 run "128 x 256: 256-row tiles" "--arrays 128 --frames 256"
run "128 x 256: 128-row tiles" "--arrays 128 --frames 256"
MCA_HIP_GEMM_BM256=1 run "fp16: 256-row tiles" "--precision fp16"
run "fp16: 128-row tiles" "--precision fp16"
