#!/bin/bash
# K steps per hipGraph at the literal batch sizes: test, then the bench leg (eager / 1-step graph / 8-step graph)
set -u
OUT=gpurun_out/r5c29; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_frame.py -x -q -m gpu --timeout 300 -k "repeatable_and_graph" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-modes --no-convert 2>$OUT/bench_$i.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d['config']['literal_batches'].items():
    print(k, 'eager', round(v['ms_per_step'],4), 'graph', v['hipgraph'], 'graph_x8', v.get('hipgraph_x8'))
"
done
