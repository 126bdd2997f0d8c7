#!/bin/bash
# Round 5, GPU call 2: the fused normalisations (md_gemm_ln_f16, md_groupnorm_table_f16 + md_gemm_affine_f16) -- parity, micro-benchmark,
# full-width UNet goldens on the fused graph, then the same-box end-to-end A/B (MD_FUSE_NORMS=0 / 1).
TAG=${1:-c2}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 420 python -m pytest tests/test_fused_norm_gpu.py -x -q > $O/pytest_fused.log 2>&1; echo "fused tests rc=$?"; tail -25 $O/pytest_fused.log
MD_ITERS=20 timeout 200 python tools/bench_kernels.py fused > $O/bench_fused.log 2>&1; grep -v amdgpu $O/bench_fused.log
timeout 900 python -m pytest tests/test_unets_gpu.py tests/test_blocks_gpu.py tests/test_full_size_gpu.py tests/test_e2e_parity_gpu.py -x -q > $O/pytest_unets.log 2>&1; echo "unets rc=$?"; tail -6 $O/pytest_unets.log
for r in 1 2; do for f in 0 1; do
  MD_FUSE_NORMS=$f timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['kernel_families']
print('== MD_FUSE_NORMS=$f (round $r): %.3f f/s  %.1f ms  gemm %.0f  groupnorm %.1f  layernorm %.1f' % (d['value'], d['ms_per_step'], f['gemm']['ms_per_clip'], f['groupnorm']['ms_per_clip'], f['layernorm']['ms_per_clip']), ' | '.join('%s %.1f' % (s['label'].replace('gemm ',''), s['ms_per_clip']) for s in d['top_launch_shapes'] if ' ln' in s['label'] or ' gn' in s['label'] or 'stats' in s['label']))"
done; done 2>&1 | tee $O/ab_fuse_norms.log
MD_BENCH_DUMP=$O/shapes_fused.txt timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-vae --no-pmc > $O/bench_fused_dump.json 2>/dev/null
grep -E " ln| gn|stats|layernorm|groupnorm" $O/shapes_fused.txt | head -40
