mkdir -p gpurun_out/$1
python scripts/site_times.py --steps 6 > gpurun_out/$1/x2.txt 2>&1
python scripts/site_times.py --steps 6 --precision bf16 > gpurun_out/$1/bf16.txt 2>&1
