#!/bin/bash
# Round 6, GPU call 7: the streaming GEMMs' leftover CUs as extra row streams (MD_WS_EXTRA = 0 / 1 in one library): correctness (every-element race screen + operator
# tests), micro-benchmarks, end to end.
R=${GRAFT_REPO_ROOT:-.}; cd $R; O=$R/gpurun_out/c7; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fused_norm_gpu.py tests/test_gemm_sp_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/ws_diag.py 2>&1 | tail -8
for r in 1 2; do for x in 0 1; do echo "== MD_WS_EXTRA=$x (round $r)"; MD_WS_EXTRA=$x MD_ITERS=30 MD_WARM=5 timeout 400 python tools/bench_kernels.py gemm skinny fused 2>&1 | grep -v amdgpu | grep -E "gemm"; done; done > $O/kern.log 2>&1
grep -E "==|x640x640|x1920x640|x1280x640|x960x320|x2560x320|ln\+gemm|gn\+gemm|294912x640x320" $O/kern.log
for r in 1 2; do for x in 0 1; do
  MD_WS_EXTRA=$x MD_BENCH_DUMP=$O/shapes_x${x}_$r.txt timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-pmc 2>/dev/null > $O/ab_x${x}_$r.json
  python - $O/ab_x${x}_$r.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); f = d["kernel_families"]
print(sys.argv[1].split("/")[-1], "%.3f f/s %.1f ms" % (d["value"], d["ms_per_step"]), " ".join("%s %.0f" % (k, v["ms_per_clip"]) for k, v in list(f.items())[:5]))
PY
done; done 2>&1 | tee $O/ab.log
for x in 0 1; do grep -E "K=640$|N=1920 K=640|N=2560 K=320 geglu|N=960 K=320 ln|K=640 T" $O/shapes_x${x}_2.txt | head -8; echo; done
