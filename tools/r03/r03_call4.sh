#!/bin/bash
# Round 3, GPU call 4: gemm_tw_kernel (two free-running waves per SIMD): parity + same-box A/B against gemm_sp_kernel and the round-2 dispatch
TAG=${1:-r3d}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
echo "== tw parity"; MD_GEMM_TW=1 timeout 600 python tests/gemm_sp_check.py > $O/tw_check.log 2>&1; echo "tw parity rc=$?"; tail -3 $O/tw_check.log
for r in 1 2; do for v in "MD_GEMM_SP=0" "MD_GEMM_SP=1" "MD_GEMM_TW=1"; do echo "== $v (round $r)"; env $v timeout 300 python tools/bench_kernels.py conv gemm shapes 2>&1 | grep -v amdgpu | grep -v "x320x320\|x640x640\|x640x320 \|8192x8192x8192\|x1280x640 \|x1920x640 \|294912x320x640"; done; done > $O/ab_tw.log 2>&1
python - <<PY
import re,collections
rows=collections.OrderedDict(); cur=None
for l in open("$O/ab_tw.log"):
    m=re.match(r"== (\S+) \(round", l)
    if m: cur=m.group(1); continue
    m=re.match(r"(.{44})\s+([\d.]+) ms\s+([\d.]+) TFLOP", l)
    if m and cur: rows.setdefault(m.group(1).strip(), collections.defaultdict(list))[cur].append(float(m.group(3)))
print("%-46s %10s %10s %10s" % ("shape (TFLOP/s, best of 2)", "round-2", "sp", "tw"))
for k,v in rows.items():
    g=lambda n: max(v[n]) if v[n] else float("nan")
    print("%-46s %10.1f %10.1f %10.1f" % (k, g("MD_GEMM_SP=0"), g("MD_GEMM_SP=1"), g("MD_GEMM_TW=1")))
PY
