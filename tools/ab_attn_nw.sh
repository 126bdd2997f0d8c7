#!/bin/bash
# same-box A/B of attention builds (tools/ab/lib_*.so) x workgroup width (MD_ATTN_NW = waves per workgroup)
cp mikudance_amd/libmdance_hip.so /tmp/lib_keep.so
for r in 1 2; do for v in "$@"; do for nw in 4 8 16; do cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so; echo "== $v MD_ATTN_NW=$nw (round $r)"; MD_ATTN_NW=$nw python tools/bench_kernels.py attn 2>&1 | grep "D=40"; done; done; done
for v in ${PARITY:-$1}; do cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so; for nw in 8 16; do echo "== parity $v MD_ATTN_NW=$nw"; MD_ATTN_NW=$nw python -m pytest tests/test_kernels_gpu.py -q -k attention 2>&1 | tail -2; done; done
cp /tmp/lib_keep.so mikudance_amd/libmdance_hip.so
