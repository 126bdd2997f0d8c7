#!/bin/bash
# Round 3, GPU call 2: full sp parity harness, dispatch-table micro-benchmarks (MD_GEMM_SP 0 / 1, tile order), ablation builds of the sp
# main loop, SQ counters of the sp kernels, one end-to-end line with the automatic rule.
TAG=${1:-r3b}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
echo "== sp parity"; MD_GEMM_SP=1 timeout 600 python tests/gemm_sp_check.py > $O/sp_check.log 2>&1; echo "sp parity rc=$?"; tail -4 $O/sp_check.log
for d in 0 1; do echo "== MD_GEMM_SP=$d shapes small"; MD_GEMM_SP=$d timeout 300 python tools/bench_kernels.py shapes 2>&1 | grep -v amdgpu; done > $O/ab_sp_shapes.log 2>&1; cat $O/ab_sp_shapes.log
for r in 1 2; do for gm in 1 8; do echo "== MD_GEMM_SP_GROUPM=$gm (round $r)"; MD_GEMM_SP=1 MD_GEMM_SP_GROUPM=$gm timeout 300 python tools/bench_kernels.py gemm 2>&1 | grep -v amdgpu | grep -v "x320x320\|x640x640\|x640x320 "; done; done > $O/ab_sp_groupm.log 2>&1; cat $O/ab_sp_groupm.log
echo "== ablations (1 no barrier, 2 no DMA, 4 no fragment reads, 8 no vmcnt wait, 15 all)"
cp mikudance_amd/libmdance_hip.so /tmp/lib_keep_ab.so
for r in 1 2; do for v in base abl1 abl2 abl4 abl8 abl15; do cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so; echo "== $v (round $r)"; MD_GEMM_SP=1 timeout 200 python tools/bench_kernels.py conv gemm 2>&1 | grep -v amdgpu | grep "conv 32x48x48 640\|conv 32x96x96 320\|x640x2560\|geglu"; done; done > $O/ab_sp_ablation.log 2>&1
cp /tmp/lib_keep_ab.so mikudance_amd/libmdance_hip.so
cat $O/ab_sp_ablation.log
echo "== e2e"
for d in 0 2; do MD_GEMM_SP=$d timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['kernel_families']
print('MD_GEMM_SP=$d: %.3f f/s  gemm %.0f ms  conv %.0f' % (d['value'], f['gemm']['ms_per_clip'], f['conv3x3']['ms_per_clip']))"; done | tee $O/e2e.log
echo "== pmc"
cd /tmp && export TMPDIR=/tmp
export MD_ITERS=2 MD_WARM=1
MD_GEMM_SP=1 timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $O/pmc_sq -o sq -- python $R/tools/bench_kernels.py conv gemm > $O/pmc_sq.log 2>&1
MD_GEMM_SP=1 timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -o fetch -- python $R/tools/bench_kernels.py conv gemm > $O/pmc_fetch.log 2>&1
cd $R
python tools/pmc_table.py $O/pmc_sq $O/pmc_fetch > $O/pmc_table.txt 2>&1; cat $O/pmc_table.txt | head -60
rm -rf $O/pmc_sq $O/pmc_fetch
