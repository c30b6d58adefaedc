#!/bin/bash
# the driver's GPU gate N times + default bench line
set -u
TAG=${1:-r5gate}; N=${2:-1}
OUT=gpurun_out/$TAG; mkdir -p $OUT
for i in $(seq 1 $N); do
  rm -f gpurun_out/parity_report.txt
  timeout 900 python -m pytest tests -x -q -m gpu --timeout 600 > $OUT/pytest_$i.log 2>&1
  echo "pytest rc=$?" >> $OUT/pytest_$i.log
  cp gpurun_out/parity_report.txt $OUT/parity_report_$i.txt 2>/dev/null
  tail -2 $OUT/pytest_$i.log
done
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open('$OUT/bench.json'))
print('ms/step', d['ms_per_step'], 'value', d['value'])
r=d['roofline']; print('roofline', r['kernel'], r['ms_per_step'], r['frac'], [ (x['kernel'][:30], round(x['ms_per_step'],3), round(x['frac'],3)) for x in r.get('trace_rows',[])])
print('sites', [(s['group'][:12], round(s['ms_per_step'],3), round(s['frac'],3), round(s.get('frac_model_B',-1),3)) for s in r.get('sites',[])])
print('modes', {k:(round(v['ms_per_step'],3) if isinstance(v,dict) else v) for k,v in d.get('modes',{}).items() if k!='note'})
print('literal', {k:(round(v['ms_per_step'],3), round(v['hipgraph'].get('ms_per_step',-1),3)) for k,v in d['config'].get('literal_batches',{}).items()})
print('convert', {k:(round(v.get('ms', v.get('ms_per_utterance',0)),4)) for k,v in d['config'].get('convert_config4',{}).items() if isinstance(v,dict)})
print('hbm_model_B', d['step_fraction_of_rooflines'])
PY
