#!/bin/bash
# Developer tool: socket power and shader clock (rocm-smi) sampled while the default train step runs in a loop; prints the samples.
# usage (on the GPU box): bash scripts/power_probe.sh [STEPS] [extra bench.py flags]
STEPS=${1:-3000}; shift
python bench.py --steps $STEPS --warmup 20 --headline-only --no-traffic --no-cpu-baseline "$@" > /tmp/pp_bench.json 2>/dev/null &
BP=$!
sleep 6
for i in $(seq 1 16); do
  /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|sclk|mclk|fclk" | tr -s ' ' | cut -c1-90 | tr '\n' '|'
  echo
  sleep 0.7
done
wait $BP
python -c "import json; d=json.load(open('/tmp/pp_bench.json')); print('ms_per_step', round(d['ms_per_step'],4), 'steps', d['steps'])"
echo "--- idle"
sleep 3
/opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|sclk|mclk" | tr -s ' ' | cut -c1-90
