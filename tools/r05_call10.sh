#!/bin/bash
# Round 5, GPU call 10: FF-out (and every plain sp GEMM whose A exceeds the memory-side cache) walking its tile order backwards
# (MD_SP_REVERSE = 1, default) against forwards (0): parity of the GEMM tests, then same-box end to end.
TAG=${1:-c10}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gemm_sp_gpu.py tests/test_kernels_gpu.py -x -q -k "gemm or wide_k" > $O/pytest_gemm.log 2>&1; echo "gemm tests rc=$?"; tail -3 $O/pytest_gemm.log
for r in 1 2; do for f in 0 1; do
  MD_SP_REVERSE=$f MD_BENCH_DUMP=$O/shapes_rev$f.txt timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['kernel_families']
print('== MD_SP_REVERSE=$f (round $r): %.3f f/s  %.1f ms  gemm %.0f' % (d['value'], d['ms_per_step'], f['gemm']['ms_per_clip']))"
  grep "N=320 K=1280\|N=640 K=2560\|N=1280 K=5120" $O/shapes_rev$f.txt | head -4
done; done 2>&1 | tee $O/ab_sp_reverse.log
timeout 600 python -m pytest tests/test_unets_gpu.py -x -q > $O/pytest_unets.log 2>&1; echo "unets rc=$?"; tail -3 $O/pytest_unets.log
