#!/bin/bash
# round 6: the shipped library against round 5's (commit e8d4c51, built into abtest/lib_r05.so) on ONE box in alternation -- what the level
# balance of the pair transforms (csrc/pair_balance.h), the fixed-lag policy and the rest of the round cost or gained on the headline step
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() {
  python bench.py --full --steps 100 --warmup 20 --cpu-frames 0 --single-stream 0 --extras 0 2> /dev/null | grep "^{" | tail -1 > /tmp/ab.json
  python - "$1" <<PY
import json,sys
d=json.load(open('/tmp/ab.json'))
print('%-28s %6.2f M frames/s  %.4f ms  ' % (sys.argv[1], d['value']/1e6, d['ms_per_step']), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items() if v['launches']})
PY
}
for rep in 1 2 3 4; do
MCA_HIP_LIB=$GRAFT_REPO_ROOT/abtest/lib_r05.so run "round 5 (e8d4c51)"
run "round 6 (shipped)"
done
