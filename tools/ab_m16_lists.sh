#!/bin/bash
# round 6: 16 microphones, ADAPTIVE -- two work lists (k_scan_pick<PL, 2>: the frames that take whole rows, i.e. the eager
# tails and the unsure rows, through k_srp_gemm_repair + k_repair_patch; every other flagged frame through k_srp_cand at its candidate columns)
# (MCA_HIP_ADAPT_CAND=1) against whole rows for every flagged frame (the default for such contexts)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
echo "--- whole rows for every flagged frame (default):"; python tools/bench_shapes.py m16a 2>&1 | grep -v amdgpu.ids
echo "--- two work lists (MCA_HIP_ADAPT_CAND=1):"; MCA_HIP_ADAPT_CAND=1 python tools/bench_shapes.py m16a 2>&1 | grep -v amdgpu.ids
done
