#!/bin/bash
# End-of-round extras: configs[4] on the final build, attention two-q-tiles-per-wave at 2 waves/SIMD (A/B build), and the
# power / clock question (the same GEMM / conv / attention kernels on all-zero operands).
TAG=${1:-r02x2}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python bench.py --config 4 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "cfg4 rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench_cfg4.json").read().strip().splitlines()[-1]); print("cfg4", round(d["value"],3), "ms/step", round(d["ms_per_step"],1), "peak GB", round(d["peak_hbm_gb"],1), "mfma frac", round(d["mfma_frac_whole_loop"],3))
PY
cp mikudance_amd/libmdance_hip.so /tmp/keep.so
for r in 1 2; do
  echo "== final, QT=1 (round $r)"; python tools/bench_kernels.py attn 2>&1 | grep "D=40"
  cp tools/ab/lib_qt2w2.so mikudance_amd/libmdance_hip.so
  echo "== two q-tiles per wave, 2 waves/SIMD (round $r)"; MD_ATTN_QT=2 python tools/bench_kernels.py attn 2>&1 | grep "D=40"
  cp /tmp/keep.so mikudance_amd/libmdance_hip.so
done > $O/ab_attn_qt2.log 2>&1; cat $O/ab_attn_qt2.log
for z in "" 1; do echo "== MD_BENCH_ZERO=$z"; MD_BENCH_ZERO=$z python tools/bench_kernels.py gemm conv attn 2>&1 | grep -v amdgpu; done > $O/zero_vs_random.log 2>&1; cat $O/zero_vs_random.log
