#!/bin/bash
# developer: interleaved A/B of variant libraries (scripts/build_variant.sh) on the 32768-frame step
# usage: scripts/ab_libs.sh [rounds] name1 name2 ...   (name "default" = the in-tree library)
R=$1; shift
for i in $(seq $R); do
  for n in "$@"; do
    if [ "$n" = default ]; then L=""; else L="variants/$n/libvaenpvc_hip.so"; fi
    VAENPVC_LIB=$L python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-literal --no-modes --no-convert --no-traffic 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n', round(d['ms_per_step'],4))"
  done
done
