#!/bin/bash
# round 5: candidate columns (shipped: k_srp_cand writes the exact values of the listed rows at the columns the flagged frames need)
# against whole rows (MCA_HIP_ADAPT_CAND=0: k_srp_gemm_repair + k_repair_patch, round 4); the shipped library
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cand in 0 1 0 1; do
  MCA_HIP_ADAPT_CAND=$cand python bench.py --full --steps 100 --warmup 20 --cpu-frames 0 2> /dev/null | grep "^{" | tail -1 > /tmp/ab.json
  python - $cand <<PY
import json,sys
d=json.load(open('/tmp/ab.json'))
print('cand %s  headline %6.2f M frames/s  %.4f ms  ' % (sys.argv[1], d['value']/1e6, d['ms_per_step']), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items() if v['launches']}, ' repair rows', d['repair']['recomputed_fraction'])
s=d['config']['single_stream_4096']; print('        1 x 4096 per call: %.4f ms (graph replay %.4f)' % (s['ms_per_call'], s['graph_replay']['ms_per_call']))
for sp in d['repair_spread']: print('        %-78s %.4f ms per 32768 frames  %.3f x  repair %.3f ms  recomputed %.4f  columns per flagged frame %.1f  whole-row frames %d' % (sp['input'][:78], sp['ms_per_32768_frames'], sp['vs_headline'], sp['repair_ms'], sp['recomputed_fraction'], sp['columns_per_flagged_frame'], sp['whole_row_frames']))
PY
done
