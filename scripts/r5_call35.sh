#!/bin/bash
# k_fbwd for decoder layer 2: register prefetch of dy (pfw1) / of the pre-LN tensor (pfw2) during the GEMMs, now that the staged values are
# no longer touched right behind their loads; the better variant then runs the parity tests of the fused kernels
set -u
OUT=gpurun_out/r5c35; mkdir -p $OUT
T="dec2_bwd"
for i in 1 2; do
  python scripts/site_times.py --tags $T > $OUT/def_$i.txt 2>&1
  VAENPVC_LIB=variants/pfw1/libvaenpvc_hip.so python scripts/site_times.py --tags $T > $OUT/pfw1_$i.txt 2>&1
  VAENPVC_LIB=variants/pfw2/libvaenpvc_hip.so python scripts/site_times.py --tags $T > $OUT/pfw2_$i.txt 2>&1
done
python scripts/cmp_sites.py $OUT/def_1.txt $OUT/pfw1_1.txt $OUT/pfw2_1.txt $OUT/def_2.txt $OUT/pfw1_2.txt $OUT/pfw2_2.txt
BEST=$(python - <<PY
import re
def t(f):
    for l in open(f):
        m=re.match(r'dec2_bwd\s+([0-9.]+) us', l)
        if m: return float(m.group(1))
    return 1e9
r={n:(t('$OUT/%s_1.txt'%n)+t('$OUT/%s_2.txt'%n))/2 for n in ('def','pfw1','pfw2')}
b=min(r,key=r.get)
print(b if (b!='def' and r[b] < r['def']-5) else 'def')
PY
)
echo "best: $BEST"
if [ "$BEST" != def ]; then
  VAENPVC_LIB=variants/$BEST/libvaenpvc_hip.so timeout 100 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 90 -k "fused or fixture or gradients or ragged_large" > $OUT/pytest.log 2>&1
  echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
fi
