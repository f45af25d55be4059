cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "MCA_HIP_ADAPT_TAU_SCALE=1" "MCA_HIP_ADAPT_TAU_SCALE=0.000001" "MCA_HIP_ADAPT_LAZY=0"; do
  export $v
  python bench.py --steps 60 --warmup 20 --cpu-frames 0 --single-stream 0 --extras 0 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,2), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items() if v['launches']}, d['repair']['flagged'])"
  unset MCA_HIP_ADAPT_TAU_SCALE MCA_HIP_ADAPT_LAZY
done
python bench.py --steps 60 --warmup 20 --cpu-frames 0 --single-stream 0 --extras 0 --precision fp16 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fp16', round(d['value']/1e6,2), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items() if v['launches']})"
