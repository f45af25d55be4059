#!/bin/bash
# round 5: what do the loads of k_beamform_wave cost? (MEASURE build; MCA_HIP_BFW_ABL 1: no sample loads, 2: no steering-row loads, 3: neither -- wrong audio on purpose)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export MCA_HIP_LIB=$GRAFT_REPO_ROOT/abtest/lib_measure.so
for abl in 0 1 2 3 0; do
  MCA_HIP_BFW_ABL=$abl python bench.py --steps 100 --warmup 20 --cpu-frames 0 --single-stream 0 --extras 0 2> /dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ABL $abl  %.4f ms  ' % d['ms_per_step'], {k: round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items() if v['launches']})"
done
