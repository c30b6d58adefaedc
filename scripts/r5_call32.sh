#!/bin/bash
# thirds staging with unconditional constant-offset loads (decoder layer 0 forward had 12 spilled offset pairs, each reloaded behind vmcnt(0)):
# parity, then same-box A/B against the previous code (variants/old)
set -u
OUT=gpurun_out/r5c32; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 600 > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
T="dec0_fwd,dec0_dgrad,enc2_fwd,enc2_dgrad,enc2_wgrad,enc3_fwd,dec1_fwd"
for i in 1 2; do
  VAENPVC_LIB=variants/old/libvaenpvc_hip.so python scripts/site_times.py --tags $T > $OUT/old_$i.txt 2>&1
  python scripts/site_times.py --tags $T > $OUT/new_$i.txt 2>&1
done
python scripts/cmp_sites.py $OUT/old_1.txt $OUT/new_1.txt $OUT/old_2.txt $OUT/new_2.txt
bash scripts/ab_libs.sh 3 old default | tee $OUT/ab.txt
for i in 1 2; do
  for L in variants/old/libvaenpvc_hip.so ""; do
    VAENPVC_LIB=$L python bench.py --steps 40 --warmup 10 --precision bf16 --no-cpu-baseline --no-literal --no-modes --no-convert 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bf16 lib=[$L]', round(d['ms_per_step'],4))"
  done
done
