#!/bin/bash
# Round 5, GPU call 5: the simplified fused normalisations (PRO_AFF + PRO_LNF, K = 320) and the many-slab GroupNorm: parity, micro-benchmarks,
# same-box end to end, VAE / temporal-VAE timing.
TAG=${1:-c5}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_fused_norm_gpu.py tests/test_kernels_gpu.py tests/test_vae_gpu.py -x -q > $O/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -5 $O/pytest_kernels.log
MD_ITERS=30 MD_WARM=5 timeout 200 python tools/bench_kernels.py fused 2>&1 | grep -v amdgpu | tee $O/bench_fused.log
timeout 600 python -m pytest tests/test_unets_gpu.py tests/test_blocks_gpu.py tests/test_e2e_parity_gpu.py -x -q > $O/pytest_unets.log 2>&1; echo "unets rc=$?"; tail -4 $O/pytest_unets.log
for r in 1 2; do for f in 0 1; do
  MD_FUSE_NORMS=$f timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['kernel_families']
print('== MD_FUSE_NORMS=$f (round $r): %.3f f/s  %.1f ms  gemm %.0f  groupnorm %.1f  layernorm %.1f' % (d['value'], d['ms_per_step'], f['gemm']['ms_per_clip'], f['groupnorm']['ms_per_clip'], f['layernorm']['ms_per_clip']))"
done; done 2>&1 | tee $O/ab_fuse_norms.log
timeout 300 python tools/bench_vae_temporal.py > $O/vae_temporal.json 2> $O/vae_temporal.err; cat $O/vae_temporal.json
MD_VAE_DUMP=$O/vae_shapes.txt timeout 300 python tools/bench_vae.py > $O/vae.json 2> $O/vae.err; cat $O/vae.json
