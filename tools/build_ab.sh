#!/bin/bash
# diagnostic / A-B builds of the library: [SRC=unit] tools/build_ab.sh NAME [extra hipcc flags for that unit, default gemm.hip]  ->  tools/ab/lib_NAME.so
# (the other objects are taken from the regular build; run `make -C mikudance_amd/csrc` first)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../mikudance_amd/csrc"
mkdir -p ../../tools/ab
SRC=${SRC:-gemm}        # which translation unit gets the extra flags (SRC=attention tools/build_ab.sh nosmallt -DA2_NO_SMALLT)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c $SRC.hip -o /tmp/${SRC}_$NAME.o
OBJS=""
for o in gemm attention norm temporal elementwise; do if [ $o = $SRC ]; then OBJS="$OBJS /tmp/${SRC}_$NAME.o"; else OBJS="$OBJS $o.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/lib_$NAME.so $OBJS
echo built tools/ab/lib_$NAME.so
