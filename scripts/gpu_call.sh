#!/bin/bash
# One gpurun call: GPU test-suite, default bench, kernel traces (32768 and 256 frames).  Everything lands in gpurun_out/$TAG.
# usage (on the GPU box, from the repo root): scripts/gpu_call.sh <tag> [pytest-args...]
set -u
TAG=${1:-c1}; shift || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rm -f $ROOT/gpurun_out/parity_report.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 "$@" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
cp $ROOT/gpurun_out/parity_report.txt $OUT/ 2>/dev/null
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?" >> $OUT/bench.err
trace() {  # name, bench args
  local name=$1; shift
  rm -rf /tmp/rp_$name
  (cd /tmp && VAENPVC_SIDE_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/rp_$name -- python $ROOT/bench.py --no-cpu-baseline --no-literal --no-modes --no-convert "$@" > $OUT/$name.log 2>&1)
  local db=$(find /tmp/rp_$name -name '*.db' | head -1)
  if [ -n "$db" ]; then python $ROOT/scripts/rocpd_stats.py $db 80 > $OUT/trace_$name.txt; else echo "no db for $name" > $OUT/trace_$name.txt; fi
}
trace big --steps 10 --warmup 3
trace f256 --frames 256 --steps 100 --warmup 10
trace f16 --frames 16 --steps 100 --warmup 10
tail -3 $OUT/pytest.log; head -c 600 $OUT/bench.json; tail -2 $OUT/bench.err
