#!/bin/bash
# Quick GPU pass: kernel parity tests, temporal flavours A/B (matrix-core vs lane-per-query), norm micro-benchmarks, one bench line.
TAG=${1:-chk2}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -m pytest tests/test_kernels_gpu.py -m gpu -q > $O/pytest_kernels.log 2>&1; echo "kernel tests rc=$?"; tail -5 $O/pytest_kernels.log
MD_TEMPORAL_MFMA=0 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k temporal > $O/pytest_temporal_old.log 2>&1; echo "temporal (lane-per-query) rc=$?"; tail -2 $O/pytest_temporal_old.log
for r in 1 2; do
  for q in 1 0; do echo "== MD_TEMPORAL_MFMA=$q (round $r)"; MD_TEMPORAL_MFMA=$q python tools/bench_kernels.py temporal norm 2>&1 | grep -v amdgpu; done
done > $O/ab_temporal.log 2>&1; cat $O/ab_temporal.log
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); f=d["kernel_families"]
print("bench %.3f f/s" % d["value"], {k: round(v["ms_per_clip"], 1) for k, v in f.items()})
PY
