#!/bin/bash
# Round 5, GPU call 9: the transposed-output (V^T) projections by kernel / tile: automatic dispatch, pinned tiles, and
# the small-tile kernel with its LDS-staged transposed store.
TAG=${1:-c9}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export MD_ITERS=30 MD_WARM=5
{
for spec in "auto:" "sp192x256:MD_GEMM_SP_NT=4" "sp128x256:MD_GEMM_SP_NT=2" "gemm_kernel_128x128:MD_GEMM_SP=0" "auto:"; do
  name=${spec%%:*}; envs=${spec#*:}
  echo "== $name"; env $envs timeout 200 python tools/bench_kernels.py tgemm 2>&1 | grep -v amdgpu
done
} | tee $O/ab_transposed_out.log
cd /tmp && export TMPDIR=/tmp MD_ITERS=3 MD_WARM=1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python $R/tools/bench_kernels.py tgemm > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc_write -o w -- python $R/tools/bench_kernels.py tgemm > /dev/null 2>&1
cd $R
{ python tools/pmc_raw.py $O/pmc_fetch --match gemm; python tools/pmc_raw.py $O/pmc_write --match gemm; } | tee -a $O/ab_transposed_out.log
rm -rf $O/pmc_fetch $O/pmc_write
