#!/bin/bash
# Round 3, GPU call 13: new dispatch thresholds + ws GEGLU with side-by-side GELU chains: full GPU suite, A/B, default bench line
TAG=${1:-r3m}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
cp mikudance_amd/libmdance_hip.so /tmp/lib_keep_ab.so
for r in 1 2; do for v in base wsg; do cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so; echo "== $v (round $r)"; timeout 300 python tools/bench_kernels.py gemm 2>&1 | grep -v amdgpu | grep "geglu"; done; done 2>&1 | tee $O/ab_wsg.log
for v in base wsg; do
  cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so
  echo "== e2e $v"
  timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vae 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['kernel_families']
print('%.3f f/s  ' % d['value'] + '  '.join('%s %.0f' % (k, v['ms_per_clip']) for k, v in f.items()))"
done 2>&1 | tee $O/e2e.log
cp /tmp/lib_keep_ab.so mikudance_amd/libmdance_hip.so
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real; tail -c 1500 $O/bench_default.json
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -5 $O/pytest_gpu.log
