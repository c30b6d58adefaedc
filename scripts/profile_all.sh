#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: produces the text summaries that
# are committed under profiles/ (kernel trace + three separate PMC passes + default bench + per-site table).
# usage: scripts/profile_all.sh <round-tag> [trace|step|all]   (step: only the large-batch train step: trace, PMC passes, site table, default bench)
set -u
TAG=${1:-r02}
WHAT=${2:-all}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# the weight-gradient stream is serialised (VAENPVC_SIDE_STREAM=0) so that a kernel's duration is not inflated
# by kernels running next to it
CMD="python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-literal --no-modes --no-convert --no-traffic"
run_pass() {  # name, parser, rocprof args...
  local name=$1 parser=$2; shift 2
  rm -rf /tmp/rp_$name
  (cd /tmp && VAENPVC_SIDE_STREAM=0 timeout 600 rocprofv3 "$@" -d /tmp/rp_$name -- $CMD > $OUT/$name.log 2>&1)
  local db=$(find /tmp/rp_$name -name '*.db' | head -1)
  if [ -n "$db" ]; then python $ROOT/scripts/$parser $db 70 > $OUT/${TAG}_$name.txt; else echo "no db for $name" > $OUT/${TAG}_$name.txt; fi
}
run_pass kernel_trace_stats rocpd_stats.py --kernel-trace --stats
if [ "$WHAT" = step ]; then
  run_pass pmc_sq rocpd_pmc.py --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  run_pass pmc_fetch_lds rocpd_pmc.py --kernel-trace --pmc FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
  run_pass pmc_write rocpd_pmc.py --kernel-trace --pmc WRITE_SIZE
  (cd $ROOT && timeout 300 python scripts/site_times.py > $OUT/${TAG}_site_times.txt 2>/dev/null)
  (cd $ROOT && timeout 900 python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err)
fi
if [ "$WHAT" = all ]; then
  run_pass pmc_sq rocpd_pmc.py --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  run_pass pmc_fetch_lds rocpd_pmc.py --kernel-trace --pmc FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
  run_pass pmc_write rocpd_pmc.py --kernel-trace --pmc WRITE_SIZE
  (cd $ROOT && timeout 300 python scripts/site_times.py > $OUT/${TAG}_site_times.txt 2>/dev/null)
  (cd $ROOT && timeout 300 python scripts/site_times.py --frames 256 > $OUT/${TAG}_site_times_F256.txt 2>/dev/null)
  (cd /tmp && VAENPVC_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_f256 -- python $ROOT/bench.py --frames 256 --steps 100 --warmup 10 --no-cpu-baseline --no-literal --no-modes --no-convert --no-traffic > $OUT/f256.log 2>&1)
  db=$(find /tmp/rp_f256 -name '*.db' | head -1); [ -n "$db" ] && python $ROOT/scripts/rocpd_stats.py $db 70 > $OUT/${TAG}_kernel_trace_stats_F256.txt
  (cd /tmp && VAENPVC_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_f16 -- python $ROOT/bench.py --frames 16 --steps 100 --warmup 10 --no-cpu-baseline --no-literal --no-modes --no-convert --no-traffic > $OUT/f16.log 2>&1)
  db=$(find /tmp/rp_f16 -name '*.db' | head -1); [ -n "$db" ] && python $ROOT/scripts/rocpd_stats.py $db 70 > $OUT/${TAG}_kernel_trace_stats_F16.txt
  # VAWGAN branch (config 5): wall times and kernel trace at 16 and 256 frames
  for VF in 16 256; do
    (cd $ROOT && timeout 300 python scripts/vawgan_bench.py --frames $VF --iters 50 2>/dev/null | tail -1 > $OUT/${TAG}_vawgan_bench_F$VF.json)
  done
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_vw -- python $ROOT/scripts/vawgan_bench.py --frames 16 --iters 20 > $OUT/vw.log 2>&1)
  db=$(find /tmp/rp_vw -name '*.db' | head -1); [ -n "$db" ] && python $ROOT/scripts/rocpd_stats.py $db 70 > $OUT/${TAG}_vawgan_kernel_trace_stats.txt
  # conversion path (config 4): encode -> decode
  (cd $ROOT && timeout 300 python scripts/bench_convert.py > $OUT/${TAG}_bench_convert.json 2> $OUT/bench_convert.err)
  # small-batch path: per-phase clocks of the two frame kernels, per-segment times of the one-launch weight gradient
  (cd $ROOT && VAENPVC_FRAME_PROF=1 timeout 300 python scripts/frame_prof.py > $OUT/${TAG}_frame_phases_F16.txt 2>/dev/null)
  (cd $ROOT && timeout 300 python scripts/wgrad_prof.py 16 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_frame_wgrad_segments_F16.txt)
  (cd $ROOT && timeout 300 python scripts/wgrad_prof.py 256 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_frame_wgrad_segments_F256.txt)
  (cd $ROOT && timeout 900 python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err)
fi
