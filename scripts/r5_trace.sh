#!/bin/bash
set -u
OUT=$(pwd)/gpurun_out/r5trace; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
rm -rf /tmp/rp_big
(cd /tmp && VAENPVC_SIDE_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/rp_big -- python $ROOT/bench.py --no-cpu-baseline --no-literal --no-modes --no-convert --steps 10 --warmup 3 > $OUT/big.log 2>&1)
db=$(find /tmp/rp_big -name '*.db' | head -1)
[ -n "$db" ] && python $ROOT/scripts/rocpd_stats.py $db 80 > $OUT/trace_big.txt
head -30 $OUT/trace_big.txt | cut -c1-150
