#!/bin/bash
set -u
OUT=gpurun_out/r5c16; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 600 -k "fused_layer_backward or ragged or benchmarked or all_tuned or unfiltered or kink or properties" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
T="dec1_bwd,lnb_dec0,dec0_wgrad,dec0_dgrad"
for i in 1 2; do
  VAENPVC_FB_LNB2=0 python scripts/site_times.py --tags $T > $OUT/l2_off_$i.txt 2>&1
  python scripts/site_times.py --tags $T > $OUT/l2_on_$i.txt 2>&1
done
python scripts/cmp_sites.py $OUT/l2_off_1.txt $OUT/l2_on_1.txt $OUT/l2_off_2.txt $OUT/l2_on_2.txt
scripts/ab_env.sh 2 "VAENPVC_FB_LNB2=0" "-" 2>&1 | tee $OUT/ab.txt
for e in "VAENPVC_FB_LNB2=0" "VAENPVC_FB_LNB2=1"; do
  env $e python bench.py --precision bf16 --steps 40 --warmup 10 --no-cpu-baseline --no-literal --no-modes --no-convert 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e bf16', round(d['ms_per_step'],4))"
done
