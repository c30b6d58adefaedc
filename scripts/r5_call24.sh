#!/bin/bash
set -u
OUT=gpurun_out/r5c24; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_frame.py -x -q -m gpu -k "twenty" 2>&1 | tail -1
grep -E 'layered-bf16x2' gpurun_out/parity_report.txt | cut -c1-180
T="dec0_fwd,merge_fwd,dec1_fwd"
for i in 1 2; do
  python scripts/site_times.py --tags $T > $OUT/clo0_$i.txt 2>&1
  VAENPVC_LIB=variants/clo1/libvaenpvc_hip.so python scripts/site_times.py --tags $T > $OUT/clo1_$i.txt 2>&1
  VAENPVC_LIB=variants/clo2/libvaenpvc_hip.so python scripts/site_times.py --tags $T > $OUT/clo2_$i.txt 2>&1
done
python scripts/cmp_sites.py $OUT/clo0_1.txt $OUT/clo1_1.txt $OUT/clo2_1.txt $OUT/clo0_2.txt $OUT/clo1_2.txt $OUT/clo2_2.txt
