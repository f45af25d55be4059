#!/bin/bash
# round 5: the candidate contraction inside the list-mode analysis launch (shipped) against k_srp_cand as a launch of its own (MCA_HIP_CAND_FUSE=0); MEASURE build
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export MCA_HIP_LIB=$GRAFT_REPO_ROOT/abtest/lib_measure.so
for f in 0 1 0 1; do
  MCA_HIP_CAND_FUSE=$f python bench.py --full --steps 100 --warmup 20 --cpu-frames 0 --single-stream 1 2> /dev/null | grep "^{" | tail -1 > /tmp/ab.json
  python - $f <<PY
import json,sys
d=json.load(open('/tmp/ab.json'))
print('fuse %s  headline %6.2f M frames/s  %.4f ms  ' % (sys.argv[1], d['value']/1e6, d['ms_per_step']), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items() if v['launches']})
s=d['config']['single_stream_4096']; print('        1 x 4096 per call: %.4f ms' % s['ms_per_call'])
for sp in d['repair_spread']: print('        %-78s %.4f ms per 32768 frames  %.3f x  repair %.3f ms' % (sp['input'][:78], sp['ms_per_32768_frames'], sp['vs_headline'], sp['repair_ms']))
PY
done
