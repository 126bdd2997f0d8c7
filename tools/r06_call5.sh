#!/bin/bash
# Round 6, GPU call 5: the residual prefetch of gemm_sp_kernel (RESP), MD_SP_PF = 0 / 1 in ONE library: correctness, trace, micro-benchmarks, end to end.
R=${GRAFT_REPO_ROOT:-.}; cd $R; O=$R/gpurun_out/c5; mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_sp_gpu.py tests/test_kernels_gpu.py tests/test_blocks_gpu.py -x -q -m gpu 2>&1 | tail -3
for pf in 0 1; do MD_SP_PF=$pf SP_TRACE_LIB=trace MD_TRACE="n1280 k640 ffout" bash tools/r06_gpu.sh c5_trace_pf$pf trace; done
for r in 1 2; do for pf in 0 1; do echo "== MD_SP_PF=$pf (round $r)"; MD_SP_PF=$pf MD_ITERS=30 MD_WARM=5 timeout 400 python tools/bench_kernels.py gemm skinny shapes conv 2>&1 | grep -v amdgpu | grep -E "gemm|conv"; done; done > $O/kern.log 2>&1
grep -E "==|\+res|294912x320x1280|18432x1280x1280|73728x640x2560|18432x1280x5120|conv 32x96x96 320|conv 32x48x48 640->640 |conv 32x24x24 1280->1280 " $O/kern.log
for r in 1 2; do for pf in 0 1; do
  MD_SP_PF=$pf MD_BENCH_DUMP=$O/shapes_pf${pf}_$r.txt timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-pmc 2>/dev/null > $O/ab_pf${pf}_$r.json
  python - $O/ab_pf${pf}_$r.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); f = d["kernel_families"]
print(sys.argv[1].split("/")[-1], "%.3f f/s %.1f ms" % (d["value"], d["ms_per_step"]), " ".join("%s %.0f" % (k, v["ms_per_clip"]) for k, v in list(f.items())[:5]))
PY
done; done 2>&1 | tee $O/ab.log
