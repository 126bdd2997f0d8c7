#!/bin/bash
# Round 5, GPU call 4: row reductions on the DPP path -- parity again, the producer's cost taken apart (piece map alone / + arithmetic / + stores),
# same-box end to end.
TAG=${1:-c4}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
cp mikudance_amd/libmdance_hip.so /tmp/lib_keep.so
timeout 420 python -m pytest tests/test_fused_norm_gpu.py -x -q > $O/pytest_fused.log 2>&1; echo "fused tests rc=$?"; tail -4 $O/pytest_fused.log
for v in base statsabl1 statsabl2 base; do
  cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so; echo "== $v"; MD_ITERS=30 MD_WARM=5 timeout 200 python tools/bench_kernels.py fused 2>&1 | grep -v amdgpu
done > $O/bench_fused.log 2>&1
cp /tmp/lib_keep.so mikudance_amd/libmdance_hip.so
cat $O/bench_fused.log
timeout 600 python -m pytest tests/test_unets_gpu.py tests/test_blocks_gpu.py tests/test_kernels_gpu.py -x -q -k "not benchmark_sizes" > $O/pytest_unets.log 2>&1; echo "unets+kernels rc=$?"; tail -4 $O/pytest_unets.log
for r in 1 2; do for f in 0 1; do
  MD_FUSE_NORMS=$f timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['kernel_families']
print('== MD_FUSE_NORMS=$f (round $r): %.3f f/s  %.1f ms  gemm %.0f  groupnorm %.1f  layernorm %.1f' % (d['value'], d['ms_per_step'], f['gemm']['ms_per_clip'], f['groupnorm']['ms_per_clip'], f['layernorm']['ms_per_clip']))"
done; done 2>&1 | tee $O/ab_fuse_norms.log
MD_BENCH_DUMP=$O/shapes_fused.txt timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-vae --no-pmc > $O/bench_fused_dump.json 2>/dev/null
grep -E " ln|gn|stats|K=320" $O/shapes_fused.txt | head -30
timeout 300 python tools/bench_vae_temporal.py > $O/vae_temporal.json 2> $O/vae_temporal.err; cat $O/vae_temporal.json
