mkdir -p gpurun_out/$1
T=enc4_fwd,heads_fwd,merge_fwd,merge_dgrad,heads_dgrad,enc4_dgrad
python scripts/site_times.py --tags $T --steps 8 > gpurun_out/$1/x2.txt 2>&1
python scripts/site_times.py --tags $T --steps 8 --precision bf16 > gpurun_out/$1/bf16.txt 2>&1
