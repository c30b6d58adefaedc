#!/bin/bash
# (variant libraries: for v in 1 2 4 8; do scripts/build_variant.sh nrabl$v "-DVAENPVC_NR_ABL=$v"; done)
# ring GEMM: main-loop ablations (variant libraries -DVAENPVC_NR_ABL=n) and PMC counters of the new loop against the old one
set -u
OUT=$(pwd)/gpurun_out/r6c7; mkdir -p $OUT
bash scripts/ab_call.sh r6c7 -t enc4_fwd,heads_fwd,enc4_dgrad -r 1 env:VAENPVC_NT_RING=1 lib:nrabl1+env:VAENPVC_NT_RING=1 lib:nrabl2+env:VAENPVC_NT_RING=1 lib:nrabl4+env:VAENPVC_NT_RING=1 lib:nrabl8+env:VAENPVC_NT_RING=1 default
export TMPDIR=/tmp; ROOT=$(pwd)
for arm in ring old; do
  if [ $arm = ring ]; then E="VAENPVC_NT_RING=1"; else E="VAENPVC_NT_RING=0"; fi
  rm -rf /tmp/rp_$arm /tmp/rp2_$arm
  (cd /tmp && env $E VAENPVC_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d /tmp/rp_$arm -- python $ROOT/scripts/site_times.py --tags enc4_fwd --steps 2 > $OUT/pmc_$arm.log 2>&1)
  db=$(find /tmp/rp_$arm -name '*.db' | head -1); [ -n "$db" ] && python $ROOT/scripts/rocpd_pmc.py $db 80 | grep -E "^kernel|k_gemm_nt" > $OUT/pmc_sq_$arm.txt
  (cd /tmp && env $E VAENPVC_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT -d /tmp/rp2_$arm -- python $ROOT/scripts/site_times.py --tags enc4_fwd --steps 2 > $OUT/pmc2_$arm.log 2>&1)
  db=$(find /tmp/rp2_$arm -name '*.db' | head -1); [ -n "$db" ] && python $ROOT/scripts/rocpd_pmc.py $db 80 | grep -E "^kernel|k_gemm_nt" > $OUT/pmc_lds_$arm.txt
  cat $OUT/pmc_sq_$arm.txt $OUT/pmc_lds_$arm.txt
done
