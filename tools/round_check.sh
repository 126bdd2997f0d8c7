#!/bin/bash
# Quick GPU pass after kernel changes: kernel-level parity tests, the wsgemm race screen, micro-benchmarks (with the temporal
# query-prefetch A/B), one end-to-end bench line, and PMC passes over the attention / temporal micro-benchmarks.
TAG=${1:-chk}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x > $O/pytest_kernels.log 2>&1; echo "kernel tests rc=$?"; tail -3 $O/pytest_kernels.log
python tools/ws_diag.py > $O/ws_diag.log 2>&1; echo "ws_diag rc=$?"; grep "TOTAL BAD" $O/ws_diag.log | sort | uniq -c
for r in 1 2; do
  for q in 1 0; do echo "== MD_TEMPORAL_QPRE=$q (round $r)"; MD_TEMPORAL_QPRE=$q python tools/bench_kernels.py temporal 2>&1 | grep -v amdgpu; done
done > $O/ab_temporal.log 2>&1; cat $O/ab_temporal.log
python tools/bench_kernels.py gemm skinny attn > $O/microbench.log 2>&1; grep -v amdgpu $O/microbench.log
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); f=d["kernel_families"]
print("bench %.3f f/s" % d["value"], {k: round(v["ms_per_clip"], 1) for k, v in f.items()})
PY
tools/pmc_ws.sh "attn temporal" $TAG/pmc > $O/pmc_summary.txt 2>&1
tools/pmc_sq_ws.sh "attn temporal" $TAG/pmc_sq > $O/pmc_sq_summary.txt 2>&1
rm -rf $O/pmc $O/pmc_sq
grep -A6 "attn2_kernel<40\|temporal_attn_kernel<16, 5" $O/pmc_summary.txt | cut -c1-160 | head -30
