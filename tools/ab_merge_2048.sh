#!/bin/bash
# round 6: the merged contraction index at 2048-sample frames (k_stft_phat_2048<..., MERGE>: 3 924 complex terms per row instead of 7 x 1 025)
# against the per-group index (MCA_HIP_NO_MERGE, MEASURE build), 8 and 4 microphones, one far-field source per array, ADAPTIVE
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export MCA_HIP_LIB=$GRAFT_REPO_ROOT/abtest/lib_measure.so
for rep in 1 2; do
echo "--- per-group index (MCA_HIP_NO_MERGE=1):"; MCA_HIP_NO_MERGE=1 python tools/bench_shapes.py n2048c 2>&1 | grep -v amdgpu.ids
echo "--- merged index (shipped):"; python tools/bench_shapes.py n2048c 2>&1 | grep -v amdgpu.ids
done
