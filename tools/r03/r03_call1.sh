#!/bin/bash
# Round 3, GPU call 1: MFMA power micro-benchmark, parity + A/B of the one-wave-per-SIMD GEMM flavour (MD_GEMM_SP=1), the new
# parity tests (2-rank DP, no-CFG golden, benchmark-size kernels, temporal F <= 32 on the matrix core), MD_ATTN_SMALL validation.
TAG=${1:-r3a}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
echo "== mfma power"; timeout 120 tools/ubench/mfma_power.bin 400000 > $O/mfma_power.log 2>&1; cat $O/mfma_power.log
echo "== sp parity"; MD_GEMM_SP=1 timeout 600 python tests/gemm_sp_check.py > $O/sp_check.log 2>&1; echo "sp parity rc=$?"; tail -5 $O/sp_check.log
for r in 1 2; do for d in 0 1; do echo "== MD_GEMM_SP=$d (round $r)"; MD_GEMM_SP=$d timeout 300 python tools/bench_kernels.py gemm conv 2>&1 | grep -v amdgpu; done; done > $O/ab_sp.log 2>&1; cat $O/ab_sp.log
echo "== new tests"
timeout 900 python -m pytest tests/test_dp_gpu.py tests/test_unets_gpu.py tests/test_kernels_gpu.py -m gpu -q -x --durations=8 \
  -k "two_ranks or g13 or single_window or benchmark_sizes or config5 or large_logits or in_place or rejects_images or temporal or groupnorm" > $O/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -15 $O/pytest_new.log
echo "== attn small"
MD_ATTN_SMALL=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention" > $O/attn_small_check.log 2>&1; echo "attn small parity rc=$?"; tail -3 $O/attn_small_check.log
for r in 1 2; do for d in 0 1; do echo "== MD_ATTN_SMALL=$d (round $r)"; MD_ATTN_SMALL=$d timeout 200 python tools/bench_kernels.py xattn 2>&1 | grep -v amdgpu; done; done > $O/ab_attn_small.log 2>&1; cat $O/ab_attn_small.log
