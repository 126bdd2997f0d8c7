#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MD_ITERS=2 MD_WARM=1
OUT=$R/gpurun_out/${2:-pmc_sq_ws}
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/sq -o sq -- python $R/tools/bench_kernels.py ${1:-skinny} > $OUT.sq.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $OUT/sq2 -o sq2 -- python $R/tools/bench_kernels.py ${1:-skinny} > $OUT.sq2.log 2>&1
python $R/tools/pmc_summary.py $OUT
