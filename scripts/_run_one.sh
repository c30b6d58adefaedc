mkdir -p gpurun_out/$1
python scripts/site_times.py --tags enc0_fwd,enc0_wgrad --steps 6 > gpurun_out/$1/x2.txt 2>&1
