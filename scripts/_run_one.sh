mkdir -p gpurun_out/$1
T=enc2_fwd,dec0_fwd,dec0_dgrad,enc2_dgrad,dec0_gsplit,dec0_asplit,dec0_wgrad
python scripts/site_times.py --tags $T --steps 6 > gpurun_out/$1/x2.txt 2>&1
VAENPVC_FCR_SITES=0 python scripts/site_times.py --tags $T --steps 6 > gpurun_out/$1/x2_off.txt 2>&1
python scripts/site_times.py --tags $T --steps 6 --precision bf16 > gpurun_out/$1/bf16.txt 2>&1
VAENPVC_FCR_SITES=0 python scripts/site_times.py --tags $T --steps 6 --precision bf16 > gpurun_out/$1/bf16_off.txt 2>&1
