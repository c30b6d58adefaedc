#!/bin/bash
# usage: scripts/ablate.sh <timer-tag> ; prints the tagged kernel's average ms for each ablation switch
for d in 0 1 2 4 8 3; do
  VAENPVC_DBG=$d python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-literal --timer-tag $1 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dbg=$d', '$1', 'avg_kernel_ms=%.4f' % r['roofline']['avg_kernel_ms'], 'step_ms=%.2f' % r['ms_per_step'])"
done
