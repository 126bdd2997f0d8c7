#!/bin/bash
# SQ / FETCH / WRITE counters of the temporal-attention micro-benchmark (three separate --pmc passes, kernel-trace only) -> gpurun_out/$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-pmc_temporal}; mkdir -p $O
export MD_ITERS=2 MD_WARM=1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $O/sq -o sq -- python $R/tools/bench_kernels.py temporal > $O/sq.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch -o fetch -- python $R/tools/bench_kernels.py temporal > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write -o write -- python $R/tools/bench_kernels.py temporal > $O/write.log 2>&1
cd $R; python tools/pmc_table.py $O/sq $O/fetch $O/write --match temporal > $O/table.md 2>&1; cat $O/table.md; python tools/pmc_raw.py $O/fetch $O/write --match temporal 2>/dev/null | head -20
