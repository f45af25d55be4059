#!/bin/bash
# A/B of the run queue of k_stft_phat_wave (round 5): needs abtest/lib_measure.so (tools/ab_build.sh measure "-DMCA_MEASURE").
# usage (GPU box, repo root): bash tools/ab_dyn.sh [arrays frames]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
A=${1:-8}; F=${2:-4096}
export MCA_HIP_LIB=$GRAFT_REPO_ROOT/abtest/lib_measure.so
run() {
  python bench.py --arrays $A --frames $F --steps 100 --warmup 20 --cpu-frames 0 --single-stream 0 --extras 0 2> /dev/null | grep "^{" | tail -1 > /tmp/ab_dyn.json
  python - "$1" <<PY
import json,sys
d=json.load(open('/tmp/ab_dyn.json'))
print('%-28s %6.2f M frames/s  %.4f ms  ' % (sys.argv[1], d['value']/1e6, d['ms_per_step']), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items() if v['launches']})
PY
}
for rep in 1 2; do
run "static runs (round 4)"
MCA_HIP_DYN=1 run "queue, first runs auto"
MCA_HIP_DYN=1 MCA_HIP_DYN_LEN0=16 run "queue, first runs 16"
MCA_HIP_DYN=1 MCA_HIP_DYN_LEN0=4 run "queue, first runs 4"
MCA_HIP_DYN=1 MCA_HIP_DYN_LEN0=2 run "queue, first runs 2"
done
