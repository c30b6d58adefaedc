#!/bin/bash
# round-6 baseline call: the driver's gate once, the default bench line (with the in-run traffic passes), per-site table of both
# fp32-class precisions, kernel trace of the 3-term mode (which rows dominate at the reference's precision)
set -u
OUT=gpurun_out/${1:-r6base}; mkdir -p $OUT
bash scripts/r5_gate.sh ${1:-r6base} 1
python scripts/site_times.py > $OUT/sites_x2.txt 2>&1; tail -1 $OUT/sites_x2.txt
python scripts/site_times.py --precision bf16x3 > $OUT/sites_x3.txt 2>&1; tail -1 $OUT/sites_x3.txt
export TMPDIR=/tmp; ROOT=$(pwd)
rm -rf /tmp/rp_x3
(cd /tmp && VAENPVC_SIDE_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/rp_x3 -- python $ROOT/bench.py --precision bf16x3 --headline-only --no-traffic --steps 8 --warmup 2 > $OUT/x3.log 2>&1)
db=$(find /tmp/rp_x3 -name '*.db' | head -1)
[ -n "$db" ] && python $ROOT/scripts/rocpd_stats.py $db 60 > $OUT/r06_kernel_trace_stats_bf16x3.txt
head -25 $OUT/r06_kernel_trace_stats_bf16x3.txt | cut -c1-140
