#!/bin/bash
# round 5: where the time of the run queue goes -- flat queues (every run the same length) and the waves' exit clocks
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export MCA_HIP_LIB=$GRAFT_REPO_ROOT/abtest/lib_measure.so
run() {
  python bench.py --steps 60 --warmup 15 --cpu-frames 0 --single-stream 0 --extras 0 2> /dev/null | grep "^{" | tail -1 > /tmp/ab_dyn.json
  python - "$1" <<PY
import json,sys
d=json.load(open('/tmp/ab_dyn.json'))
print('%-34s %6.2f M frames/s  %.4f ms  ' % (sys.argv[1], d['value']/1e6, d['ms_per_step']), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items() if v['launches']})
PY
}
clock() {   # exit-clock spread of the waves of the last launch (100 MHz wall clock)
  python - "$1" <<PY
import sys, numpy as np
a=np.loadtxt(sys.argv[1]).reshape(-1,4)
a=a[a[:,2]>0]
t0=a[:,1].min(); end=(a[:,2]-t0)/100.0; start=(a[:,1]-t0)/100.0
print('   waves %d  start spread %.1f us  exit: min %.1f  10%% %.1f  median %.1f  90%% %.1f  max %.1f us   mean exit %.1f   runs per wave min %d max %d' % (
      len(a), start.max(), end.min(), np.percentile(end,10), np.median(end), np.percentile(end,90), end.max(), end.mean(), a[:,3].min(), a[:,3].max()))
PY
}
MCA_HIP_WAVE_CLOCK=/tmp/wc_static.txt run "static runs of 16"; clock /tmp/wc_static.txt
for L in 16 8 4 2 1; do MCA_HIP_DYN=1 MCA_HIP_DYN_FLAT=1 MCA_HIP_DYN_LEN0=$L MCA_HIP_WAVE_CLOCK=/tmp/wc_$L.txt run "queue, every run $L frames"; clock /tmp/wc_$L.txt; done
MCA_HIP_DYN=1 MCA_HIP_WAVE_CLOCK=/tmp/wc_auto.txt run "queue, 8 4 2 1"; clock /tmp/wc_auto.txt
