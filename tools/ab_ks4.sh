#!/bin/bash
# round 5: 16 384-row contractions on 256 x 384 tiles: four K quarters (256 workgroups, shipped) against two halves (128 workgroups; MCA_HIP_GEMM_KS2, MEASURE build)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export MCA_HIP_LIB=$GRAFT_REPO_ROOT/abtest/lib_measure.so
for rep in 1 2; do
echo "-- two K halves (round 4)"; MCA_HIP_GEMM_KS2=1 python tools/bench_shapes.py m16 2>&1 | grep "^M="
echo "-- four K quarters"; python tools/bench_shapes.py m16 2>&1 | grep "^M="
done
echo "-- 8 microphones, 4 arrays x 4096 frames (16 384 rows), adaptive: two halves / four quarters"
for v in 1 0; do
  if [ $v = 1 ]; then export MCA_HIP_GEMM_KS2=1; else unset MCA_HIP_GEMM_KS2; fi
  python bench.py --arrays 4 --steps 100 --warmup 20 --cpu-frames 0 --single-stream 0 --extras 0 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  %.2f M frames/s  %.4f ms  ' % (d['value']/1e6, d['ms_per_step']), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items() if v['launches']})"
done
