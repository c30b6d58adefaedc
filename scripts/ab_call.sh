#!/bin/bash
# ONE gpurun call = one experiment (replaces round 5's 35 one-off r5_call*.sh scripts): an optional parity subset, an interleaved same-box A/B of
# kernel-site times, an optional interleaved whole-step A/B -- for any mix of variant libraries and environment switches.
#
#   scripts/ab_call.sh OUT  -t TAGS  [-k PYTEST_EXPR] [-r ROUNDS] [-s STEP_ROUNDS] [-p PRECISION] [-f FRAMES]  ARM [ARM ...]
#
#   ARM    "default" | "lib:NAME" (variants/NAME/libvaenpvc_hip.so, built by scripts/build_variant.sh NAME "-D...") | "env:A=1,B=0"
#          | "lib:NAME+env:A=1" (both)
#   -t     comma separated site tags of scripts/site_times.py ("" = skip the site table)
#   -k     pytest -k expression run FIRST on every arm that is not "default" (parity before speed); "" = none
#   -r     rounds of the interleaved site table (default 2);  -s  rounds of the interleaved 40-step bench (default 0 = none)
#   -m     bench mode flags appended to the step A/B (e.g. "--precision bf16x3")
# Output: gpurun_out/OUT/{arm}_{i}.txt, a side-by-side table (scripts/cmp_sites.py) and the step times on stdout.
set -u
OUT=gpurun_out/$1; shift
TAGS=""; KEXPR=""; R=2; SR=0; PREC=auto; FR=32768; BMODE=""
while getopts "t:k:r:s:p:f:m:" o; do
  case $o in t) TAGS=$OPTARG;; k) KEXPR=$OPTARG;; r) R=$OPTARG;; s) SR=$OPTARG;; p) PREC=$OPTARG;; f) FR=$OPTARG;; m) BMODE=$OPTARG;; esac
done
shift $((OPTIND - 1))
mkdir -p $OUT
arm_env() {   # prints the env assignments of an arm
  local a=$1 e=""
  IFS='+' read -ra parts <<< "$a"
  for p in "${parts[@]}"; do
    case $p in
      default) ;;
      lib:*) e="$e VAENPVC_LIB=variants/${p#lib:}/libvaenpvc_hip.so";;
      env:*) e="$e $(echo ${p#env:} | tr ',' ' ')";;
    esac
  done
  echo $e
}
arm_name() { echo "$1" | tr -c 'A-Za-z0-9_=\n' '_' | cut -c1-40; }
if [ -n "$KEXPR" ]; then
  for a in "$@"; do
    [ "$a" = default ] && continue
    n=$(arm_name "$a")
    env $(arm_env "$a") timeout 600 python -m pytest tests -x -q -m gpu --timeout 300 -k "$KEXPR" > $OUT/pytest_$n.log 2>&1
    echo "pytest[$a] rc=$? $(tail -1 $OUT/pytest_$n.log)"
  done
fi
if [ -n "$TAGS" ]; then
  files=""
  for i in $(seq $R); do
    for a in "$@"; do
      n=$(arm_name "$a")
      env $(arm_env "$a") python scripts/site_times.py --tags $TAGS --precision $PREC --frames $FR > $OUT/${n}_$i.txt 2>&1
      files="$files $OUT/${n}_$i.txt"
    done
  done
  python scripts/cmp_sites.py $files
fi
for i in $(seq $SR); do
  for a in "$@"; do
    env $(arm_env "$a") python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-literal --no-modes --no-convert --no-traffic $BMODE 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step[$a]', round(d['ms_per_step'],4))"
  done
done
