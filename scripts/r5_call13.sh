#!/bin/bash
set -u
OUT=gpurun_out/r5c13; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 600 -k "fused_thin or ragged or benchmarked or all_tuned or unfiltered or view_conv or encode_decode" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
T="enc2_fwd,enc3_split,enc3_fwd"
for i in 1 2; do
  VAENPVC_E2_OSP=0 python scripts/site_times.py --tags $T > $OUT/osp_off_$i.txt 2>&1
  python scripts/site_times.py --tags $T > $OUT/osp_on_$i.txt 2>&1
done
python scripts/cmp_sites.py $OUT/osp_off_1.txt $OUT/osp_on_1.txt $OUT/osp_off_2.txt $OUT/osp_on_2.txt
scripts/ab_env.sh 2 "VAENPVC_E2_OSP=0" "-" 2>&1 | tee $OUT/ab.txt
