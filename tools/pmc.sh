#!/bin/bash
# PMC passes (separate runs, kernel-trace only) over the kernel micro-benchmarks; summaries land in gpurun_out/pmc_*/
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MD_ITERS=2 MD_WARM=1
WHAT="${1:-gemm conv attn}"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_sq -o sq -- python $R/tools/bench_kernels.py $WHAT > $R/gpurun_out/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o fetch -- python $R/tools/bench_kernels.py $WHAT > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/pmc_write -o write -- python $R/tools/bench_kernels.py $WHAT > $R/gpurun_out/pmc_write.log 2>&1
ls -la $R/gpurun_out/pmc_sq $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
