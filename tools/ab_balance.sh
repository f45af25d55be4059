#!/bin/bash
# round 6: the level balance of a transform pair (csrc/pair_balance.h) -- what it costs on balanced input (the bench streams) and what
# it repairs (tests/test_gpu_pair_levels.py with it switched off: MCA_HIP_NO_BALANCE, MEASURE build)
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
MCA_HIP_NO_BALANCE=1 run "no balance (round 5)"
run "balance (shipped)"
done
echo "--- tests/test_gpu_pair_levels.py with the balance switched off (round 5's analysis): expected to FAIL"
MCA_HIP_NO_BALANCE=1 timeout 600 python -m pytest tests/test_gpu_pair_levels.py -q -x --timeout 500 2>&1 | tail -5
