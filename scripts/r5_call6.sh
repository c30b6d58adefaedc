#!/bin/bash
# round 5, call 6: result tile of C = A B^T through LDS (16-byte stores): parity + A/B
set -u
OUT=gpurun_out/r5c6; mkdir -p $OUT
rm -f gpurun_out/parity_report.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 600 -k "a_resident or plane_gemm or benchmarked or unfiltered or properties or ragged or all_tuned" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
T="enc4_fwd,heads_fwd,merge_fwd,merge_dgrad,heads_dgrad,enc4_dgrad"
for i in 1 2; do
  VAENPVC_NT_LEP=0 VAENPVC_NT_AR=0 python scripts/site_times.py --tags $T > $OUT/lep_off_$i.txt 2>&1
  VAENPVC_NT_AR=0 python scripts/site_times.py --tags $T > $OUT/lep_on_$i.txt 2>&1
done
python scripts/cmp_sites.py $OUT/lep_off_1.txt $OUT/lep_on_1.txt $OUT/lep_off_2.txt $OUT/lep_on_2.txt
scripts/ab_env.sh 2 "VAENPVC_NT_LEP=0" "-" 2>&1 | tee $OUT/ab.txt
