#!/bin/bash
TAG=${1:-r3h}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
cp mikudance_amd/libmdance_hip.so /tmp/lib_keep_ab.so
for r in 1 2; do for v in base abl64 abl2; do cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so; echo "== $v (round $r)"; MD_GEMM_SP=1 timeout 200 python tools/bench_kernels.py gemm 2>&1 | grep -v amdgpu | grep "x640x2560\|10240x1280 geglu\|8192 geglu\|x1280x5120\|5120x640 geglu"; done; done 2>&1 | tee $O/ab_sp_abl64.log
cp /tmp/lib_keep_ab.so mikudance_amd/libmdance_hip.so
