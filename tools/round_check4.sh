#!/bin/bash
TAG=${1:-chk4}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for r in 1 2; do for z in 0 1; do for mb in 0 24 48 96 144; do echo "== MD_GN_CHUNK_MB=$mb MD_GN_ZIGZAG=$z (round $r)"; MD_GN_ZIGZAG=$z MD_GN_CHUNK_MB=$mb python tools/bench_kernels.py norm 2>&1 | grep groupnorm; done; done; done > $O/ab_gn.log 2>&1; cat $O/ab_gn.log
MD_GN_ZIGZAG=1 MD_GN_CHUNK_MB=48 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "norm" 2>&1 | tail -2
