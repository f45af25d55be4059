#!/bin/bash
# round 5: the older workgroup of every CU (dispatched first, favoured by the CU's oldest-first issue) takes more frames per wave
# (MCA_HIP_SPW_SKEW: the analysis kernel, MCA_HIP_BFW_SKEW: the beamformer; MEASURE build; 0 = off, unset = the shipped rule)
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
for rep in 1 2; do
MCA_HIP_SPW_SKEW=0 MCA_HIP_BFW_SKEW=0 run "no skew (round 4)"
MCA_HIP_BFW_SKEW=0 run "analysis 19 / 13, beamformer even"
for s in 1 2 3 4 5; do MCA_HIP_BFW_SKEW=$s run "analysis 19 / 13, beamformer ft +- $s"; done
done
MCA_HIP_SPW_SKEW=0 MCA_HIP_BFW_SKEW=0 run "128 x 256: no skew" "--arrays 128 --frames 256"
run "128 x 256: shipped rule" "--arrays 128 --frames 256"
