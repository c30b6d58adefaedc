mkdir -p gpurun_out/$1
python scripts/site_times.py --tags dec3_wgrad,dec3_fwd,dec3_dgrad --steps 8 > gpurun_out/$1/base.txt 2>&1
VAENPVC_LIB=variants/dg2/libvaenpvc_hip.so python scripts/site_times.py --tags dec3_wgrad,dec3_fwd,dec3_dgrad --steps 8 > gpurun_out/$1/dg2.txt 2>&1
VAENPVC_LIB=variants/dg2/libvaenpvc_hip.so python scripts/site_times.py --tags dec3_wgrad,dec3_fwd,dec3_dgrad --steps 8 --precision bf16 > gpurun_out/$1/dg2_bf16.txt 2>&1
python scripts/site_times.py --tags dec3_wgrad,dec3_fwd,dec3_dgrad --steps 8 --precision bf16 > gpurun_out/$1/base_bf16.txt 2>&1
