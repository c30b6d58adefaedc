#!/bin/bash
# d(h) as planes out of decoder layer 0's input-gradient kernel: parity, then same-box A/B against the split pass
set -u
OUT=gpurun_out/r5c27; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_runtime.py -x -q -m gpu --timeout 600 > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
T="dec0_dgrad,merge_dsplit,merge_segsum,merge_wgrad,merge_dgrad"
for i in 1 2; do
  python scripts/site_times.py --tags $T > $OUT/on_$i.txt 2>&1
  VAENPVC_D0G_PLANES=0 python scripts/site_times.py --tags $T > $OUT/off_$i.txt 2>&1
done
python scripts/cmp_sites.py $OUT/off_1.txt $OUT/on_1.txt $OUT/off_2.txt $OUT/on_2.txt
bash scripts/ab_env.sh 3 "VAENPVC_D0G_PLANES=0" "VAENPVC_D0G_PLANES=1" | tee $OUT/ab.txt
