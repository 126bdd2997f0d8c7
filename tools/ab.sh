#!/bin/bash
# same-box A/B of two (or more) builds of libmdance_hip.so (tools/ab/lib_*.so) on the kernel micro-benchmarks
cp mikudance_amd/libmdance_hip.so /tmp/lib_keep_ab.so
for r in 1 2; do for v in "$@"; do cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so; echo "== $v (round $r)"; python tools/bench_kernels.py ${WHAT:-gemm} 2>&1 | grep -v amdgpu | grep "${FILTER:-.}"; done; done
cp /tmp/lib_keep_ab.so mikudance_amd/libmdance_hip.so
