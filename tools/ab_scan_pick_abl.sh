#!/bin/bash
# round 5: where k_scan_pick's time goes -- the shipped library against a build whose peak pick is compiled out (-DMCA_ABL_PICK: wrong results), adaptive (MODE 1) and fp16 (MODE 0)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lib in "" "$GRAFT_REPO_ROOT/abtest/lib_ablpick.so"; do
  for prec in adaptive fp16; do
    env ${lib:+MCA_HIP_LIB=$lib} MCA_HIP_ADAPT_TAU_SCALE=0.000001 python bench.py --steps 60 --warmup 20 --cpu-frames 0 --single-stream 0 --extras 0 --precision $prec 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-12s %-9s' % ('${lib:+no pick}', '$prec'), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items() if v['launches']})"
  done
done
