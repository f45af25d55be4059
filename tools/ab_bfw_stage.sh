#!/bin/bash
# round 6 (the one bounded attempt on the headline kernels): k_beamform_wave with the next step's samples staged through
# LDS by global_load_lds_dwordx4 (MCA_HIP_BFW_VAR=31, MEASURE build) against the shipped variant (15); same bits required
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export MCA_HIP_LIB=$GRAFT_REPO_ROOT/abtest/lib_measure.so
run() {
  python bench.py --full --steps 100 --warmup 20 --cpu-frames 0 --single-stream 0 --extras 0 $2 2> /dev/null | grep "^{" | tail -1 > /tmp/ab.json
  python - "$1" <<PY
import json,sys
d=json.load(open('/tmp/ab.json'))
print('%-44s %6.2f M frames/s  %.4f ms  ' % (sys.argv[1], d['value']/1e6, d['ms_per_step']), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items() if v['launches']})
PY
}
for rep in 1 2 3; do
MCA_HIP_BFW_VAR=15 run "shipped (registers)"
MCA_HIP_BFW_VAR=31 run "staged through LDS"
done
echo "--- same bits?"
python - <<'PY'
import os, numpy as np, torch
from mcarray_amd import api, synth
import bench
dev = torch.device("cuda:0")
pcm = bench.synth_batch(synth.ULA8, [0x5EED0000 + a for a in range(8)], 512, dev)[0]
outs = []
for var in ("15", "31"):
    os.environ["MCA_HIP_BFW_VAR"] = var
    ctx = api.Context(48000, synth.ULA8, 1024, 0.5, 1, srp_precision=api.SRP_FP16X3, max_arrays=8)
    b = torch.empty(8, 512, 1, dtype=torch.int32, device=dev); r = torch.empty(8, 512, 1, dtype=torch.float32, device=dev)
    q = torch.empty(8, 512, 1, dtype=torch.float32, device=dev); o = torch.empty(8, 1, 512 * 512, dtype=torch.float32, device=dev)
    ctx.process_frames_dev(pcm[:, :, :513 * 512].contiguous(), 512, b, r, q, None, o); torch.cuda.synchronize()
    outs.append(o.cpu().numpy()); ctx.close()
print("audio bit-identical:", np.array_equal(outs[0], outs[1]), "max |difference|", float(np.abs(outs[0] - outs[1]).max()), "max |audio|", float(np.abs(outs[0]).max()))
PY
