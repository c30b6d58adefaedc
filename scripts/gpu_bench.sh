#!/bin/bash
# bench + kernel trace only.  usage: scripts/gpu_bench.sh <tag> [extra bench args for the trace]
set -u
TAG=${1:-b}; shift || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
rm -rf /tmp/rp_big
(cd /tmp && VAENPVC_SIDE_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/rp_big -- python $ROOT/bench.py --no-cpu-baseline --no-literal --no-modes --no-convert --steps 10 --warmup 3 "$@" > $OUT/big.log 2>&1)
db=$(find /tmp/rp_big -name '*.db' | head -1)
[ -n "$db" ] && python $ROOT/scripts/rocpd_stats.py $db 80 > $OUT/trace_big.txt
python - <<PY
import json
d=json.load(open('$OUT/bench.json'))
print('ms/step', d['ms_per_step'], 'modes', {k:(round(v['ms_per_step'],3) if isinstance(v,dict) else v) for k,v in d.get('modes',{}).items() if k!='note'})
print('roofline', d['roofline']['avg_kernel_ms'], d['roofline']['frac'], {k:round(v['avg_kernel_ms'],3) for k,v in d['roofline'].get('sibling_kernels',{}).items()})
print('literal', {k:(round(v['ms_per_step'],3), round(v['hipgraph'].get('ms_per_step',-1),3)) for k,v in d['config'].get('literal_batches',{}).items()})
PY
