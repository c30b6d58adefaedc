mkdir -p gpurun_out/$1
T=dec3_fwd,dec3_dgrad,dec3_wgrad
python scripts/site_times.py --tags $T --steps 8 > gpurun_out/$1/x2.txt 2>&1
python scripts/site_times.py --tags $T --steps 8 --precision bf16 > gpurun_out/$1/bf16.txt 2>&1
python scripts/site_times.py --tags $T --steps 8 --precision bf16x3 > gpurun_out/$1/x3.txt 2>&1
