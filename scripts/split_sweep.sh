#!/bin/bash
# developer: train-step time over batch sizes (optionally VAENPVC_FRAME_SPLIT_MAX=0|512 to force the split launches off / on)
for F in ${@:-16 32 64 96 128 160 192 256}; do
  python bench.py --frames $F --steps 300 --warmup 30 --no-cpu-baseline --no-literal --no-modes --no-convert 2>/dev/null \
    | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['frames_per_step_per_gpu'], round(d['ms_per_step'],4))"
done
