#!/bin/bash
# Round 3, GPU call 5: where does the DMA cost of gemm_sp_kernel go -- matrix-pipe duty or shader clock?  SQ / GRBM counters of the
# regular build and of the ablation builds without DMA pieces (abl2) and without DMA, reads, barrier, waits (abl15).
TAG=${1:-r3e}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export MD_ITERS=3 MD_WARM=1 MD_GEMM_SP=1
cp $R/mikudance_amd/libmdance_hip.so /tmp/lib_keep_ab.so
for v in base abl2 abl15; do
  cp $R/tools/ab/lib_$v.so $R/mikudance_amd/libmdance_hip.so
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $O/sq_$v -o sq -- python $R/tools/bench_kernels.py conv > $O/pmc_$v.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_LDS_ADDR_CONFLICT -d $O/sq2_$v -o sq2 -- python $R/tools/bench_kernels.py conv > $O/pmc2_$v.log 2>&1
  echo "== $v"; python $R/tools/pmc_table.py $O/sq_$v; python $R/tools/pmc_raw.py $O/sq2_$v
done 2>&1 | tee $O/pmc_ablation.txt
cp /tmp/lib_keep_ab.so $R/mikudance_amd/libmdance_hip.so
rm -rf $O/sq_* $O/sq2_*
