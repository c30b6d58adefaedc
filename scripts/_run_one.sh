mkdir -p gpurun_out/$1
T=enc3_dgrad,dec0_dgrad,enc3_fwd
python scripts/site_times.py --tags $T --steps 6 > gpurun_out/$1/base.txt 2>&1
VAENPVC_CG_TAIL64=1 python scripts/site_times.py --tags $T --steps 6 > gpurun_out/$1/t64.txt 2>&1
python scripts/site_times.py --tags $T --steps 6 --precision bf16 > gpurun_out/$1/base_bf16.txt 2>&1
VAENPVC_CG_TAIL64=1 python scripts/site_times.py --tags $T --steps 6 --precision bf16 > gpurun_out/$1/t64_bf16.txt 2>&1
