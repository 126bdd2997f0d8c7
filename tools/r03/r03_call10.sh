#!/bin/bash
# Round 3, GPU call 10: per-shape table of the clip with the automatic dispatch, the short-K shapes on sp vs ws, SQ / fetch counters of the sp kernels
TAG=${1:-r3j}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
MD_BENCH_DUMP=$O/shapes_all.txt timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench.json 2>/dev/null; head -70 $O/shapes_all.txt
for d in 2 1; do echo "== MD_GEMM_SP=$d"; MD_GEMM_SP=$d timeout 300 python tools/bench_kernels.py skinny 2>&1 | grep -v amdgpu; done | tee $O/ab_skinny.log
cd /tmp && export TMPDIR=/tmp
export MD_ITERS=3 MD_WARM=1 MD_GEMM_SP=1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $O/pmc_sq -o sq -- python $R/tools/bench_kernels.py conv gemm > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -o fetch -- python $R/tools/bench_kernels.py conv gemm > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/pmc_write -o write -- python $R/tools/bench_kernels.py conv gemm > $O/pmc_write.log 2>&1
cd $R
python tools/pmc_table.py $O/pmc_sq $O/pmc_fetch | tee $O/pmc_table.txt
python tools/pmc_raw.py $O/pmc_write | tee -a $O/pmc_table.txt
rm -rf $O/pmc_sq $O/pmc_fetch $O/pmc_write
