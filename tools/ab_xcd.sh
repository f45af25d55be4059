#!/bin/bash
# round 5: does it matter which frames the workgroups of one XCD take? (MEASURE build, MCA_HIP_SPW_XCD)
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
run "consecutive workgroups -> consecutive XCDs"
MCA_HIP_SPW_XCD=1 run "every XCD one contiguous piece"
done
run "128 x 256: round-robin" "--arrays 128 --frames 256"
MCA_HIP_SPW_XCD=1 run "128 x 256: contiguous per XCD" "--arrays 128 --frames 256"
