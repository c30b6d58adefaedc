#!/bin/bash
set -u
OUT=gpurun_out/r5c18; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 600 -k "ragged or benchmarked or all_tuned or unfiltered or properties or encode_decode or fused_thin" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
T="dec2_fwd,dec2_stats_planes,dec3_fwd,loss"
for i in 1 2; do
  VAENPVC_D2_TAIL=0 python scripts/site_times.py --tags $T > $OUT/tail_off_$i.txt 2>&1
  python scripts/site_times.py --tags $T > $OUT/tail_on_$i.txt 2>&1
done
python scripts/cmp_sites.py $OUT/tail_off_1.txt $OUT/tail_on_1.txt $OUT/tail_off_2.txt $OUT/tail_on_2.txt
scripts/ab_env.sh 2 "VAENPVC_D2_TAIL=0" "-" 2>&1 | tee $OUT/ab.txt
