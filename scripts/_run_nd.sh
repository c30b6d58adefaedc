mkdir -p gpurun_out/ab1
B="python bench.py --no-cpu-baseline --no-literal --no-modes --steps 60 --warmup 10"
for i in 1 2; do
VAENPVC_FC_SITES=0 VAENPVC_CV_SITES=0x4 $B > gpurun_out/ab1/old_$i.json 2>/dev/null
$B > gpurun_out/ab1/new_$i.json 2>/dev/null
VAENPVC_FC_SITES=0 $B > gpurun_out/ab1/nofc_$i.json 2>/dev/null
VAENPVC_FC_SITES=0xfff $B > gpurun_out/ab1/allfc_$i.json 2>/dev/null
done
VAENPVC_FC_SITES=0 VAENPVC_CV_SITES=0x4 $B --precision bf16 > gpurun_out/ab1/old_bf16.json 2>/dev/null
$B --precision bf16 > gpurun_out/ab1/new_bf16.json 2>/dev/null
