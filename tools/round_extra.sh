#!/bin/bash
# Second GPU-box pass of a round: configs[4] (1024x1024, 48 frames, 30 steps) bench line, config-1-size latency, PMC passes
# (fabric bytes / L2 hits / SQ busy) over the kernel micro-benchmarks incl. the streaming GEMM.
TAG=${1:-r02x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -m pytest tests/test_vae_gpu.py tests/test_inference_script_gpu.py -m gpu -q > $O/pytest_new.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_new.log
python bench.py --config 4 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "cfg4 rc=$?"
python bench.py --size 256 --frames 4 --ddim-steps 4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg0_shape.json 2> $O/bench_cfg0.err; echo "cfg0-shape rc=$?"
tools/pmc_ws.sh "skinny attn" $TAG/pmc > $O/pmc_summary.txt 2>&1
tools/pmc_sq_ws.sh "skinny attn" $TAG/pmc_sq > $O/pmc_sq_summary.txt 2>&1
python - <<PY
import json
for f in ("bench_cfg4.json","bench_cfg0_shape.json"):
    try:
        d=json.loads(open("$O/"+f).read().strip().splitlines()[-1]); print(f, round(d["value"],3), "ms/step", round(d["ms_per_step"],1), "kernel ms", round(d["kernel_ms_per_clip"],1), "peak GB", round(d["peak_hbm_gb"],1), d["metric"])
    except Exception as e: print(f, "ERR", e)
PY
grep -A4 "attn2_kernel<40\|wsgemm_kernel<10, 5, false, false" $O/pmc_summary.txt | cut -c1-200 | head -30
