#!/bin/bash
# developer: interleaved A/B of environment settings on the 32768-frame step.  usage: scripts/ab_env.sh ROUNDS "NAME=VAL" "NAME=VAL" ...  ("-" = none)
R=$1; shift
for i in $(seq $R); do
  for e in "$@"; do
    if [ "$e" = "-" ]; then E=""; else E="$e"; fi
    env $E python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-literal --no-modes --no-convert --no-traffic 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e', round(d['ms_per_step'],4))"
  done
done
