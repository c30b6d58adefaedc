mkdir -p gpurun_out/$1
B="python bench.py --no-cpu-baseline --no-literal --no-modes --frames 256 --steps 300 --warmup 30"
$B > gpurun_out/$1/base.json 2>/dev/null
VAENPVC_FWD_MASK=0xefffffff VAENPVC_BWD_MASK=0xefffffff $B > gpurun_out/$1/plane.json 2>/dev/null
VAENPVC_FWD_MASK=0xfd7fffff VAENPVC_BWD_MASK=0xfc7fffff $B > gpurun_out/$1/fused.json 2>/dev/null
VAENPVC_FWD_MASK=0xed7fffff VAENPVC_BWD_MASK=0xec7fffff $B > gpurun_out/$1/both.json 2>/dev/null
VAENPVC_FWD_MASK=0xe97fffff VAENPVC_BWD_MASK=0xe87fffff $B > gpurun_out/$1/all.json 2>/dev/null
