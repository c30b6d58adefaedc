#!/bin/bash
# quick FETCH_SIZE pass for the TN kernels: $1 = lib ("" default)
ROOT=$(pwd); export TMPDIR=/tmp
rm -rf /tmp/rp_q
(cd /tmp && VAENPVC_LIB=$1 VAENPVC_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/rp_q -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-literal --no-modes --no-convert > /tmp/q.log 2>&1)
db=$(find /tmp/rp_q -name '*.db' | head -1)
python $ROOT/scripts/rocpd_pmc.py $db 70 | grep -E "k_gemm_tn|kernel  " | cut -c1-150
