#!/bin/bash
# Round 5, GPU call 8: gn_apply_kernel requests its first rows and gamma / beta before the statistics are assembled (base) against the
# round's previous form (-DGN_NO_PREFETCH), and GN_UNROLL = 8 against 4: parity, micro-benchmark, same-box end to end.
TAG=${1:-c8}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
cp mikudance_amd/libmdance_hip.so /tmp/lib_keep.so
timeout 600 python -m pytest tests/test_fused_norm_gpu.py tests/test_kernels_gpu.py tests/test_vae_gpu.py tests/test_blocks_gpu.py -x -q > $O/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 $O/pytest_kernels.log
for v in noprefetch base unroll8 noprefetch base unroll8; do
  cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so; echo "== $v"; MD_ITERS=30 MD_WARM=5 timeout 200 python tools/bench_kernels.py norm 2>&1 | grep -v amdgpu | grep "groupnorm"
done > $O/bench_norm.log 2>&1
cat $O/bench_norm.log
for r in 1 2; do for v in noprefetch base unroll8; do
  cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['kernel_families']
print('== $v (round $r): %.3f f/s  %.1f ms  gemm %.0f  groupnorm %.1f  layernorm %.1f' % (d['value'], d['ms_per_step'], f['gemm']['ms_per_clip'], f['groupnorm']['ms_per_clip'], f['layernorm']['ms_per_clip']))"
done; done 2>&1 | tee $O/ab_gn_prefetch.log
cp /tmp/lib_keep.so mikudance_amd/libmdance_hip.so
timeout 600 python -m pytest tests/test_unets_gpu.py -x -q > $O/pytest_unets.log 2>&1; echo "unets rc=$?"; tail -3 $O/pytest_unets.log
