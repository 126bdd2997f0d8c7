#!/bin/bash
# Round 3, GPU call 12: sp epilogue with look-ahead loads + stage-major GELU, 192 x 256 tile, non-temporal norm stores
TAG=${1:-r3l}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gemm_sp_gpu.py -m gpu -q -x > $O/pytest_sp.log 2>&1; echo "sp tests rc=$?"; tail -15 $O/pytest_sp.log
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "geglu or gemm or conv" > $O/pytest_k.log 2>&1; echo "kernel tests rc=$?"; tail -3 $O/pytest_k.log
cp mikudance_amd/libmdance_hip.so /tmp/lib_keep_ab.so
run() { # lib sp nt sets...
  local lib=$1 sp=$2 nt=$3; shift 3
  cp tools/ab/lib_$lib.so mikudance_amd/libmdance_hip.so
  echo "== lib=$lib sp=$sp nt=$nt"
  MD_GEMM_SP=$sp MD_GEMM_SP_NT=$nt timeout 400 python tools/bench_kernels.py "$@" 2>&1 | grep -v amdgpu
}
{
run base 1 0 gemm shapes conv small
run new 1 5 gemm shapes conv small
run new 1 4 gemm shapes conv small
run new 0 0 gemm shapes small
run new 2 0 gemm shapes conv small
run base 2 0 gemm shapes conv small
} 2>&1 | tee $O/ab_sp.log | grep -c TFLOP
{
for r in 1 2; do run new 2 0 norm; run nt 2 0 norm; done
} 2>&1 | tee $O/ab_nt.log | grep -c GB
for v in base new nt; do
  cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so
  echo "== e2e $v"
  timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vae 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['kernel_families']
print('%.3f f/s  ' % d['value'] + '  '.join('%s %.0f' % (k, v['ms_per_clip']) for k, v in f.items()))"
done 2>&1 | tee $O/e2e.log
cp /tmp/lib_keep_ab.so mikudance_amd/libmdance_hip.so
