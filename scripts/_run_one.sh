mkdir -p gpurun_out/$1
python scripts/site_times.py --tags enc1_fwd,dec1_fwd,dec2_fwd,dec2_dgrad,dec1_dgrad,enc1_dgrad,enc2_dgrad --steps 8 > gpurun_out/$1/x2.txt 2>&1
python scripts/site_times.py --tags enc1_fwd,dec1_fwd,dec2_fwd,dec2_dgrad,dec1_dgrad,enc1_dgrad,enc2_dgrad --steps 8 --precision bf16 > gpurun_out/$1/bf16.txt 2>&1
cd /tmp && export TMPDIR=/tmp && VAENPVC_SIDE_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/rp_x -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-literal --no-modes --steps 10 --warmup 3 > /dev/null 2>&1; db=$(find /tmp/rp_x -name '*.db' | head -1); python $GRAFT_REPO_ROOT/scripts/rocpd_stats.py $db 80 > $GRAFT_REPO_ROOT/gpurun_out/$1/trace.txt
