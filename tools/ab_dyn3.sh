#!/bin/bash
# round 5: structure of the exit-clock spread of the static analysis kernel (per workgroup, per XCD)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export MCA_HIP_LIB=$GRAFT_REPO_ROOT/abtest/lib_measure.so
MCA_HIP_WAVE_CLOCK=$GRAFT_REPO_ROOT/gpurun_out/r05_wave_clock_static.txt python bench.py --steps 30 --warmup 10 --cpu-frames 0 --single-stream 0 --extras 0 > /dev/null 2>&1
python - <<PY
import numpy as np
a=np.loadtxt('gpurun_out/r05_wave_clock_static.txt').reshape(-1,4)
t0=a[:,1].min(); end=(a[:,2]-t0)/100.0
wg=end.reshape(-1,4)
print('waves', len(end), 'exit mean %.1f max %.1f' % (end.mean(), end.max()))
print('within a workgroup: mean spread (max - min of its 4 waves) %.1f us; std of workgroup means %.1f us' % ((wg.max(1)-wg.min(1)).mean(), wg.mean(1).std()))
lin=np.arange(len(wg))   # linear workgroup id = by * 64 + bx
for m in (8, 2, 4, 16, 32, 64):
    g=[wg[lin % m == i].mean() for i in range(m)]
    print('mean exit by workgroup id mod %2d:' % m, ' '.join('%.0f' % x for x in g[:16]))
print('mean exit by array (blockIdx.y):', ' '.join('%.0f' % wg[i*64:(i+1)*64].mean() for i in range(8)))
print('mean exit by wave in workgroup:', ' '.join('%.0f' % wg[:,i].mean() for i in range(4)))
PY
