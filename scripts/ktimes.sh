#!/bin/bash
# usage: scripts/ktimes.sh tag1 tag2 ... ; prints each tagged kernel's average ms (F = 32768)
for t in "$@"; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-literal --timer-tag $t 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-12s avg_kernel_ms=%.4f step_ms=%.2f' % ('$t', r['roofline']['avg_kernel_ms'], r['ms_per_step']))"
done
