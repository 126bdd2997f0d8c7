#!/bin/bash
# Round 5, GPU call 1 (one box): cheap same-box A/Bs first, then the configs[4] 30-step parity record.
#   1. streaming GEMM residual stream: prefetch depth 4 (base) / 8 / 16 at K = 640, 6 at K = 320, and the two
#      ablations (loads without adds, adds without loads)
#   2. GELU with the |x| = inf guard (+1 v_min per value) against the shipped one on the GEGLU GEMMs
#   3. d = 40 self-attention with head-major Q / K (item 5): time, then FETCH_SIZE of both layouts
#   4. tests/e2e_parity.py --config4 --steps 30 (item 3a)
TAG=${1:-c1}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export MD_ITERS=30 MD_WARM=5
cp mikudance_amd/libmdance_hip.so /tmp/lib_keep.so
{
for v in base rd8 rd16 rd6all resnoadd resnoload base rd8 rd16 rd6all; do
  cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so; echo "== $v"; timeout 200 python tools/bench_kernels.py skinny 2>&1 | grep -v amdgpu
done
} > $O/ab_ws_residual.log 2>&1
export MD_ITERS=20
{
for v in base geluinf base geluinf; do
  cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so; echo "== $v"; timeout 200 python tools/bench_kernels.py gemm small 2>&1 | grep geglu
done
} > $O/ab_gelu_inf.log 2>&1
export MD_ITERS=10 MD_WARM=3
{
for r in 1 2; do
  cp tools/ab/lib_base.so mikudance_amd/libmdance_hip.so; timeout 200 python tools/probe_attn_layout.py token 2>&1 | grep -v amdgpu
  cp tools/ab/lib_headmajor.so mikudance_amd/libmdance_hip.so; timeout 200 python tools/probe_attn_layout.py head 2>&1 | grep -v amdgpu
done
} > $O/ab_attn_headmajor.log 2>&1
cd /tmp && export TMPDIR=/tmp MD_ITERS=3 MD_WARM=1
cp $R/tools/ab/lib_base.so $R/mikudance_amd/libmdance_hip.so
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_tok -o f -- python $R/tools/probe_attn_layout.py token > /dev/null 2>&1
cp $R/tools/ab/lib_headmajor.so $R/mikudance_amd/libmdance_hip.so
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_head -o f -- python $R/tools/probe_attn_layout.py head > /dev/null 2>&1
cd $R
{ echo "token-major:"; python tools/pmc_raw.py $O/pmc_tok --match attn; echo "head-major:"; python tools/pmc_raw.py $O/pmc_head --match attn; echo "(FETCH_SIZE in KB of 64-byte requests x 2 per 128-byte line on gfx950: double it)"; } >> $O/ab_attn_headmajor.log 2>&1
rm -rf $O/pmc_tok $O/pmc_head
cp /tmp/lib_keep.so mikudance_amd/libmdance_hip.so
tail -n 40 $O/ab_ws_residual.log; cat $O/ab_gelu_inf.log $O/ab_attn_headmajor.log
timeout 1150 python tests/e2e_parity.py --config4 --steps 30 --no-fp16-oracle --out $O/e2e_parity_cfg4_30steps.json > $O/e2e_parity_cfg4_30steps.log 2>&1; echo "e2e cfg4 30 steps rc=$?"
grep -v "^{" $O/e2e_parity_cfg4_30steps.log | tail -4
python - <<PY
import json
try:
    d=json.load(open("$O/e2e_parity_cfg4_30steps.json")); print("hip_vs_o32", d["hip_vs_o32"]["rel_l2"], d["hip_vs_o32"]["cosine"])
except Exception as e: print("ERR", e)
PY
