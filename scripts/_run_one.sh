mkdir -p gpurun_out/$1
T=enc1_fwd,dec1_fwd,dec2_fwd,dec1_dgrad,enc1_dgrad
python scripts/site_times.py --tags $T --steps 6 > gpurun_out/$1/tf4.txt 2>&1
VAENPVC_LIB=variants/tf2/libvaenpvc_hip.so python scripts/site_times.py --tags $T --steps 6 > gpurun_out/$1/tf2.txt 2>&1
