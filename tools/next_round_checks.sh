#!/bin/bash
# First GPU call of the next round: validate and A/B what was written after round 2's GPU budget ran out (all default-off).
#   MD_GEMM_PP_DIRECT=1  plain ping-pong epilogue straight from the accumulators (gemm_pp.h)
#   MD_ATTN_SMALL=1      cross-attention flavour with K / V^T resident in LDS (attention_v2s.h)
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/next
mkdir -p $O
cd $R
MD_GEMM_PP=1 MD_GEMM_PP_DIRECT=1 python tests/gemm_pp_check.py > $O/pp_direct_check.log 2>&1; echo "pp direct parity rc=$?"; tail -3 $O/pp_direct_check.log
for r in 1 2; do for d in 0 1; do echo "== MD_GEMM_PP_DIRECT=$d (round $r)"; MD_GEMM_PP_DIRECT=$d python tools/bench_kernels.py gemm conv 2>&1 | grep -v amdgpu; done; done > $O/ab_pp_direct.log 2>&1; cat $O/ab_pp_direct.log
MD_ATTN_SMALL=1 python -m pytest tests/test_kernels_gpu.py tests/test_unets_gpu.py -m gpu -q -x -k "attention or g4 or g8" > $O/attn_small_check.log 2>&1; echo "attn small parity rc=$?"; tail -3 $O/attn_small_check.log
for r in 1 2; do for d in 0 1; do echo "== MD_ATTN_SMALL=$d (round $r)"; MD_ATTN_SMALL=$d python tools/bench_kernels.py xattn 2>&1 | grep -v amdgpu; done; done | tee $O/ab_attn_small.log
# the full per-shape table (the bench JSON keeps only the top 16): where the transposed-output (V^T) projections and the 12x12 level sit
MD_BENCH_DUMP=$O/shapes_all.txt python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; head -60 $O/shapes_all.txt
for d in 0 1; do MD_GEMM_PP_DIRECT=$d python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['kernel_families']
print('MD_GEMM_PP_DIRECT=$d: %.3f f/s  gemm %.0f ms  conv %.0f' % (d['value'], f['gemm']['ms_per_clip'], f['conv3x3']['ms_per_clip']))"; done
