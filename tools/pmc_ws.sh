#!/bin/bash
# PMC passes over the skinny-GEMM micro-benchmarks (W-stationary streaming kernel): fabric bytes + L2 hit rate per launch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MD_ITERS=2 MD_WARM=1
OUT=$R/gpurun_out/${2:-pmc_ws}
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python $R/tools/bench_kernels.py ${1:-skinny} > $OUT.fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/write -o write -- python $R/tools/bench_kernels.py ${1:-skinny} > $OUT.write.log 2>&1
python $R/tools/pmc_summary.py $OUT
