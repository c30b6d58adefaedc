mkdir -p gpurun_out/$1
T=enc4_wgrad,merge_wgrad,heads_wgrad,enc3_wgrad,dec0_wgrad,enc2_wgrad
python scripts/site_times.py --tags $T --steps 6 > gpurun_out/$1/k32.txt 2>&1
VAENPVC_TN_K16=1 python scripts/site_times.py --tags $T --steps 6 > gpurun_out/$1/k16.txt 2>&1
python scripts/site_times.py --tags $T --steps 6 --precision bf16 > gpurun_out/$1/k32_bf16.txt 2>&1
VAENPVC_TN_K16=1 python scripts/site_times.py --tags $T --steps 6 --precision bf16 > gpurun_out/$1/k16_bf16.txt 2>&1
