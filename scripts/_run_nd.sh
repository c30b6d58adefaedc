mkdir -p gpurun_out/tn1
T=merge_wgrad,heads_wgrad,enc4_wgrad,enc3_wgrad,enc2_wgrad,dec0_wgrad
for x in 0 1; do
  VAENPVC_CV_SITES=0xe2ce VAENPVC_TN_XCD=$x python scripts/site_times.py --tags $T > gpurun_out/tn1/x2_xcd$x.txt 2>&1
  VAENPVC_TN_XCD=$x python scripts/site_times.py --tags $T --precision bf16 > gpurun_out/tn1/bf16_xcd$x.txt 2>&1
done
