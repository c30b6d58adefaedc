#!/bin/bash
set -u
OUT=gpurun_out/r5c15; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 600 -k "bf16_mode or fused_thin or view_conv" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
for i in 1 2; do
  for e in "VAENPVC_E2_OSP=0" "VAENPVC_E2_OSP=1" "VAENPVC_FW_SITES=0x3f"; do
    env $e python bench.py --precision bf16 --steps 40 --warmup 10 --no-cpu-baseline --no-literal --no-modes --no-convert 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e bf16', round(d['ms_per_step'],4))"
  done
done 2>&1 | tee $OUT/ab_bf16.txt
python scripts/site_times.py --precision bf16 > $OUT/sites_bf16.txt 2>&1; tail -1 $OUT/sites_bf16.txt
VAENPVC_FW_SITES=0x3f python scripts/site_times.py --precision bf16 > $OUT/sites_bf16_fw.txt 2>&1; tail -1 $OUT/sites_bf16_fw.txt
