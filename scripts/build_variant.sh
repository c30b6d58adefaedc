#!/bin/bash
# Developer tool: build a variant library with extra -D flags into variants/<name>/libvaenpvc_hip.so
# (select it with VAENPVC_LIB=variants/<name>/libvaenpvc_hip.so).  usage: scripts/build_variant.sh NAME "-DFOO=1 ..."
set -e
NAME=$1; FLAGS=$2
ROOT=$(cd $(dirname $0)/.. && pwd)
OUT=$ROOT/variants/$NAME
mkdir -p $OUT
cd $ROOT/vae-npvc_amd/csrc
for f in abi.hip runtime.hip generic_kernels.hip misc_kernels.hip disc.hip gfx950_*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $FLAGS -c $f -o $OUT/${f%.hip}.o &
done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -x hip -c model.cpp -o $OUT/model.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libvaenpvc_hip.so $OUT/*.o
echo built $OUT/libvaenpvc_hip.so
