#!/bin/bash
# diagnostic / A-B builds of the library: tools/build_ab.sh NAME [extra hipcc flags for gemm.hip]  ->  tools/ab/lib_NAME.so
# (the other objects are taken from the regular build; run `make -C mikudance_amd/csrc` first)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../mikudance_amd/csrc"
mkdir -p ../../tools/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c gemm.hip -o /tmp/gemm_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/lib_$NAME.so /tmp/gemm_$NAME.o attention.o norm.o temporal.o elementwise.o
echo built tools/ab/lib_$NAME.so
