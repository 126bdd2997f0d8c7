#!/bin/bash
# Round 3, GPU call 3: staggered DMA issue slots (A/B against the unstaggered build), parity harnesses, the full GPU suite after the
# knob / variant clean-up, end-to-end lines with the round-2 dispatch and the automatic rule.
TAG=${1:-r3c}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
echo "== sp parity"; MD_GEMM_SP=1 timeout 600 python tests/gemm_sp_check.py > $O/sp_check.log 2>&1; echo "sp parity rc=$?"; tail -2 $O/sp_check.log
cp mikudance_amd/libmdance_hip.so /tmp/lib_keep_ab.so
for r in 1 2; do for v in nostagger base; do cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so; echo "== $v (round $r)"; MD_GEMM_SP=1 timeout 200 python tools/bench_kernels.py conv gemm 2>&1 | grep -v amdgpu | grep -v "x320x320\|x640x640\|x640x320 \|8192x8192x8192"; done; done > $O/ab_sp_stagger.log 2>&1
cp /tmp/lib_keep_ab.so mikudance_amd/libmdance_hip.so
cat $O/ab_sp_stagger.log
echo "== auto rule on the shape table"; MD_GEMM_SP=2 timeout 300 python tools/bench_kernels.py shapes gemm conv 2>&1 | grep -v amdgpu > $O/shapes_auto.log; cat $O/shapes_auto.log
echo "== e2e"
for d in 0 2; do MD_GEMM_SP=$d timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['kernel_families']
print('MD_GEMM_SP=$d: %.3f f/s  ' % d['value'] + '  '.join('%s %.0f' % (k, v['ms_per_clip']) for k, v in f.items()))"; done | tee $O/e2e.log
echo "== full gpu suite"
timeout 1200 python -m pytest tests -m gpu -q -x --durations=10 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -16 $O/pytest.log
