#!/bin/bash
# Round 3, GPU call 11: cheaper exact GELU (A&S 7.1.28 form on packed fp32) in every GEGLU epilogue: parity + A/B + end to end
TAG=${1:-r3k}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_gemm_pp_gpu.py tests/test_gemm_sp_gpu.py -m gpu -q -x -k "geglu or gemm_pp or gemm_sp or streaming" > $O/pytest_geglu.log 2>&1; echo "geglu tests rc=$?"; tail -3 $O/pytest_geglu.log
cp mikudance_amd/libmdance_hip.so /tmp/lib_keep_ab.so
for r in 1 2; do for v in bk64 gelu2; do cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so; echo "== $v (round $r)"; timeout 300 python tools/bench_kernels.py gemm shapes 2>&1 | grep -v amdgpu | grep geglu; done; done 2>&1 | tee $O/ab_gelu.log
cp /tmp/lib_keep_ab.so mikudance_amd/libmdance_hip.so
echo "== e2e"
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['kernel_families']
print('%.3f f/s  ' % d['value'] + '  '.join('%s %.0f' % (k, v['ms_per_clip']) for k, v in f.items()))" | tee $O/e2e.log
