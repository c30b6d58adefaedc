#!/bin/bash
# round-6 evidence set in ONE call: kernel trace + PMC passes + site table + default bench line (profile_all step), the 3-term trace,
# the site table of the round-5 kernel selection beside this round's (same box)
set -u
bash scripts/profile_all.sh r06 step
OUT=$(pwd)/gpurun_out/prof_r06; ROOT=$(pwd); export TMPDIR=/tmp
rm -rf /tmp/rp_x3
(cd /tmp && VAENPVC_SIDE_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/rp_x3 -- python $ROOT/bench.py --precision bf16x3 --headline-only --no-traffic --steps 8 --warmup 2 > $OUT/x3.log 2>&1)
db=$(find /tmp/rp_x3 -name '*.db' | head -1); [ -n "$db" ] && python $ROOT/scripts/rocpd_stats.py $db 70 > $OUT/r06_kernel_trace_stats_bf16x3.txt
VAENPVC_NT_RING=0 VAENPVC_CG_SF_RING=0 VAENPVC_D2_LNA=0 timeout 300 python scripts/site_times.py > $OUT/r06_site_times_round5_selection.txt 2>/dev/null
bash scripts/ab_call.sh r6final_ab -t "" -s 3 env:VAENPVC_NT_RING=0,VAENPVC_CG_SF_RING=0,VAENPVC_D2_LNA=0 default > $OUT/r06_step_ab_vs_round5_selection.txt 2>&1
cat $OUT/r06_step_ab_vs_round5_selection.txt
tail -1 $OUT/r06_site_times.txt; tail -1 $OUT/r06_site_times_round5_selection.txt
python -c "
import json; d=json.load(open('$OUT/r06_bench_default.json')); print(d['ms_per_step'], d['value'], d['step_traffic']['hbm_bytes_per_step'], d['step_traffic']['measured_in_run'], d['roofline']['kernel'], d['roofline']['frac'], {k:v['ms_per_step'] for k,v in d['modes'].items() if k!='note'})"
