#!/bin/bash
# round 5: one workgroup of eight waves per CU (MCA_HIP_SPW_WAVES=8) against two of four with / without the skewed run lengths (MEASURE build)
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
MCA_HIP_SPW_SKEW=0 run "2 x 4 waves, 16 / 16 frames"
run "2 x 4 waves, 19 / 13 frames (shipped)"
MCA_HIP_SPW_WAVES=8 run "1 x 8 waves, 16 frames"
done
