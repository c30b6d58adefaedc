#!/bin/bash
# round 5, call 7: A-resident merge GEMM with the LDS epilogue vs the one-tile kernel with it
set -u
OUT=gpurun_out/r5c7; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 600 -k "a_resident or benchmarked or unfiltered or properties" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
T="merge_fwd,dec0_fwd"
for i in 1 2 3; do
  VAENPVC_NT_AR=0 python scripts/site_times.py --tags $T > $OUT/ar_off_$i.txt 2>&1
  python scripts/site_times.py --tags $T > $OUT/ar_on_$i.txt 2>&1
done
python scripts/cmp_sites.py $OUT/ar_off_1.txt $OUT/ar_on_1.txt $OUT/ar_off_2.txt $OUT/ar_on_2.txt $OUT/ar_off_3.txt $OUT/ar_on_3.txt
scripts/ab_env.sh 2 "VAENPVC_NT_AR=0" "-" 2>&1 | tee $OUT/ab.txt
