#!/bin/bash
TAG=${1:-chk3}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "norm or temporal" > $O/pytest_kernels.log 2>&1; echo "kernel tests rc=$?"; tail -3 $O/pytest_kernels.log
for r in 1 2; do
  for kb in 48 24 32 64 90; do echo "== MD_TEMPORAL_LDS_KB=$kb (round $r)"; MD_TEMPORAL_LDS_KB=$kb python tools/bench_kernels.py temporal 2>&1 | grep -v amdgpu; done
done > $O/ab_temporal_lds.log 2>&1; cat $O/ab_temporal_lds.log
python tools/bench_kernels.py norm 2>&1 | grep -v amdgpu
MD_TEMPORAL_LDS_KB=90 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "temporal" 2>&1 | tail -2
