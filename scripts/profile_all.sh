#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: produces the text summaries that
# are committed under profiles/ (kernel trace + three separate PMC passes + default bench).
# usage: scripts/profile_all.sh <round-tag> [trace|all]
set -u
TAG=${1:-r01}
WHAT=${2:-all}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-literal"
run_pass() {  # name, parser, rocprof args...
  local name=$1 parser=$2; shift 2
  rm -rf /tmp/rp_$name
  (cd /tmp && timeout 600 rocprofv3 "$@" -d /tmp/rp_$name -- $CMD > $OUT/$name.log 2>&1)
  local db=$(find /tmp/rp_$name -name '*.db' | head -1)
  if [ -n "$db" ]; then python $ROOT/scripts/$parser $db 60 > $OUT/${TAG}_$name.txt; else echo "no db for $name" > $OUT/${TAG}_$name.txt; fi
}
run_pass kernel_trace_stats rocpd_stats.py --kernel-trace --stats
if [ "$WHAT" = all ]; then
  run_pass pmc_sq rocpd_pmc.py --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  run_pass pmc_fetch_lds rocpd_pmc.py --kernel-trace --pmc FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
  run_pass pmc_write rocpd_pmc.py --kernel-trace --pmc WRITE_SIZE
  (cd $ROOT && timeout 900 python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err)
fi
