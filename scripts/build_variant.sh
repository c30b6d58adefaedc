#!/bin/bash
# usage: scripts/build_variant.sh NAME "-DVAENPVC_ABL=1 ..."   -> variants/NAME/libvaenpvc_hip.so
# (kernel experiments: the layer file is rebuilt with extra flags and linked with the regular
#  objects; select with VAENPVC_LIB=variants/NAME/libvaenpvc_hip.so)
set -e
NAME=$1; FLAGS=$2
cd "$(dirname "$0")/../vae-npvc_amd/csrc"
make -s
mkdir -p ../../variants/$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result $FLAGS -c gfx950_layers.hip -o ../../variants/$NAME/gfx950_layers.o
OBJS=$(ls *.o | grep -v gfx950_layers.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/$NAME/libvaenpvc_hip.so $OBJS ../../variants/$NAME/gfx950_layers.o
echo built variants/$NAME
