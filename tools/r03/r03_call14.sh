#!/bin/bash
# Round 3, GPU call 14: 128 x 256 sp tile for the 12 x 12 level (M = 4608): parity, A/B against the 192-row tiles, end to end
TAG=${1:-r3n}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gemm_sp_gpu.py -m gpu -q -x > $O/pytest_sp.log 2>&1; echo "sp tests rc=$?"; tail -15 $O/pytest_sp.log
run() { local sp=$1 nt=$2; shift 2; echo "== sp=$sp nt=$nt"; MD_GEMM_SP=$sp MD_GEMM_SP_NT=$nt timeout 400 python tools/bench_kernels.py "$@" 2>&1 | grep -v amdgpu; }
{
run 1 4 shapes small
run 1 2 shapes small
run 2 0 shapes small
run 1 4 shapes small
run 1 2 shapes small
} 2>&1 | tee $O/ab_t24.log | grep -c TFLOP
cp mikudance_amd/libmdance_hip.so /tmp/lib_keep_ab.so
for v in base t24; do
  cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so
  echo "== e2e $v"
  timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vae 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['kernel_families']
print('%.3f f/s  ' % d['value'] + '  '.join('%s %.0f' % (k, v['ms_per_clip']) for k, v in f.items()))"
done 2>&1 | tee $O/e2e.log
cp /tmp/lib_keep_ab.so mikudance_amd/libmdance_hip.so
